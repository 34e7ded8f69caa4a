"""Solver-relevant flags of the reference's global ``Settings`` class.

Mirrors ``config.py:7-170`` of the reference for exactly the attributes that reach the
ST ("MPC") hot path (``config.py:30-37, 94-110, 143, 145-154``); everything else in the
reference's ``Settings`` (RL training, logging, SUMO) is outside this package's scope.
``load_from_file`` keeps the reference's semantics (``config.py:161-170``): every key of
the JSON file is set as a class attribute, so the shipped experiment configs
(``configs/*.json``) load unchanged.
"""
import json


class Settings:
    # Simulation (config.py:30-37)
    TICK_LENGTH = 0.2
    MAX_POSITIVE_ACCELERATION = 4.5
    MAX_NEGATIVE_ACCELERATION = -6.0
    MINIMUM_NEGATIVE_JERK = -5.0
    MAXIMUM_POSITIVE_JERK = 5.0
    MAX_SPEED = 30
    CAR_LENGTH = 5.0

    # S-T solver and continuous reward (config.py:94)
    DESIRED_SPEED = 30.0

    # S-T solver (config.py:97-110)
    USE_CYTHON = True            # kept for config compatibility; this package always runs the HIP solver
    USE_FAST_ST_SOLVER = True
    S_DISCRETIZATION = 0.05
    T_DISCRETIZATION = 0.30
    FUTURE_S = 150.0
    FUTURE_T = 5.0
    START_UNCERTAINTY = 0.0
    UNCERTAINTY_PER_SECOND = 0.0
    V_WEIGHT = 0.5
    A_WEIGHT = 10.0
    J_WEIGHT = 10.0
    D_WEIGHT = 10.0
    MIN_ALLOWED_DISTANCE = 5
    CRASH_MIN_S = 12
    MERGE_POINT_X = -50            # config.py:34: the episode runner records the "disruption" columns past this s (control.py:289)

    # The policy's state vector, dqn.get_state_vector_from_base_state (config.py:48-49, 136-139)
    SENSOR_RADIUS = 125
    USE_ACCELERATION_OF_OTHER_CARS = True
    CARS_AHEAD = 2
    CARS_BEHIND = 2
    USE_SPEED_DIFFERENCE = True
    NORMALIZE_VECTOR_INPUT = True

    # Prediction (config.py:143)
    MAX_PREDICTED_DECELERATION = -4

    # Combined controller (config.py:146-155)
    ROLLOUT_LENGTH = 5
    ST_TEST_ROLLOUTS = 5
    LIMIT_DQN_SPEED = False
    TEST_ST_STRICTLY_BETTER = True
    TEST_ROLLOUT_STATE = True
    CHECK_ROLLOUT_CRASH = True
    COMBINATION_MIN_DISTANCE = 5.1
    STOP_X = 65
    REMEMBER_LAST_CHOICE_FOR_SWITCHING_COMBINED = False

    @classmethod
    def export_settings(cls):
        """All flags as a dict (the reference's ``Settings.export_settings``, config.py:157-159)."""
        return {name: value for name, value in vars(cls).items()
                if not name.startswith("_") and not isinstance(value, (classmethod, staticmethod)) and not callable(value)}

    @classmethod
    def load_from_file(cls, filename):
        """Every key of the JSON file becomes a class attribute (config.py:161-170); JSON objects get integer keys, as the
        reference's per-level tables need."""
        with open(filename, "r", encoding="utf-8") as fh:
            for key, value in json.load(fh).items():
                setattr(cls, key, {int(k): v for k, v in value.items()} if isinstance(value, dict) else value)

    @classmethod
    def snapshot(cls):
        """Copy of the current solver flags (used by the suite to restore state)."""
        return dict(cls.export_settings())

    @classmethod
    def restore(cls, snap):
        for k, v in snap.items():
            setattr(cls, k, v)


#: The synthetic benchmark point of BASELINE.json (H=40, A=21, K=6), mapped onto the
#: reference's parameters as SURVEY.md section 8(d) prescribes: 40 time layers of 0.3 s,
#: FUTURE_S = 360 m (S = 7201 cells), an acceleration-limited window of
#: (a_max - a_min) * dt^2 / ds = 20.16 cells (fan-out 20-21) with non-binding jerk limits.
SYNTHETIC_H40A21 = {
    "T_DISCRETIZATION": 0.30,
    "FUTURE_T": 11.7,
    "S_DISCRETIZATION": 0.05,
    "FUTURE_S": 360.0,
    "MAX_POSITIVE_ACCELERATION": 5.2,
    "MAX_NEGATIVE_ACCELERATION": -6.0,
    "MINIMUM_NEGATIVE_JERK": -35.0,
    "MAXIMUM_POSITIVE_JERK": 35.0,
    "MAX_SPEED": 30,
    "DESIRED_SPEED": 30.0,
    "V_WEIGHT": 0.5, "A_WEIGHT": 10.0, "J_WEIGHT": 10.0, "D_WEIGHT": 10.0,
    "MIN_ALLOWED_DISTANCE": 5, "CRASH_MIN_S": 20,
    "START_UNCERTAINTY": 0.0, "UNCERTAINTY_PER_SECOND": 0.0,
}

#: Solver parameters shared by all 86 shipped configs (e.g. configs/st_low.json:14-25).
REFERENCE_DEFAULT = {
    "S_DISCRETIZATION": 0.05, "T_DISCRETIZATION": 0.30, "FUTURE_S": 150.0, "FUTURE_T": 5.0,
    "START_UNCERTAINTY": 0.0, "UNCERTAINTY_PER_SECOND": 0.0,
    "V_WEIGHT": 0.5, "A_WEIGHT": 10.0, "J_WEIGHT": 10.0, "D_WEIGHT": 10.0,
    "MIN_ALLOWED_DISTANCE": 5, "CRASH_MIN_S": 20,
    "MAX_POSITIVE_ACCELERATION": 4.5, "MAX_NEGATIVE_ACCELERATION": -6.0,
    "MINIMUM_NEGATIVE_JERK": -5.0, "MAXIMUM_POSITIVE_JERK": 5.0,
    "MAX_SPEED": 30, "DESIRED_SPEED": 30.0,
}


def apply_overrides(overrides):
    """Set ``Settings`` attributes from a dict (same effect as ``load_from_file`` on a JSON)."""
    for k, v in overrides.items():
        setattr(Settings, k, v)
