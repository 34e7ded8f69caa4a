"""s-coordinate maps of the reference's ``control.py`` that the ST path uses.

Only the pure geometry helpers are mirrored (``control.py:37-38, 366-389``); the TraCI
wrappers and the SUMO episode runner of the reference's ``control.py`` are out of scope.
``set_ego_speed`` is a hook: the reference's ``st.do_st_control`` side-effects the
simulator through it (``st.py:776,782``); here it records the command and forwards it to
an optional callback so a caller-owned simulator can be attached.
"""
import math

merge_point = (-50.9, 1.72)
merge_point2 = (1.5, -1.5)
merge_point3 = (-51, -1.5)


def distance(point1, point2):
    # control.py:37-38 (float ** int is a libm pow() call in CPython)
    return math.sqrt((point1[0] - point2[0]) ** 2 + (point1[1] - point2[1]) ** 2)


merge_distance = distance(merge_point, merge_point2)
common_s = merge_point2[0] - merge_point3[0]


def get_ego_s(ego_position):
    # control.py:373-380
    ego_x, ego_y = ego_position
    if ego_x < merge_point[0]:
        return - distance(ego_position, merge_point)
    elif ego_x < merge_point2[0]:
        return distance(ego_position, merge_point)
    else:
        return ego_x - merge_point2[0] + common_s


def get_obstacle_s(vehicle_position):
    # control.py:383-385
    return vehicle_position[0] - merge_point3[0]


def get_obstacle_s_from_x(vehicle_x):
    # control.py:388-389
    return vehicle_x - merge_point3[0]


_speed_sink = None
last_commanded_speed = None


def attach_speed_sink(callback):
    """Register ``callback(speed)`` to receive what the reference would send to TraCI."""
    global _speed_sink
    _speed_sink = callback


def set_ego_speed(speed):
    # control.py:174-176 minus the TraCI call
    global last_commanded_speed
    last_commanded_speed = speed
    if _speed_sink is not None:
        _speed_sink(speed)
