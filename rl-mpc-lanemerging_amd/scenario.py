"""Geometry of the reference's SUMO network that the batched episodes need (scenario DATA: lane centre lines of merge.net.xml, metres).

``RAMP_LANE_X / RAMP_LANE_Y``: the 104 points of lane ``ramp_0`` (merge.net.xml:52), from the ramp's start to the junction entry
(-50.58, 1.71), 201.91 m along the polyline; the junction's internal lane ``:mergenode_1_0`` continues in a straight line to
(1.50, -1.60) (merge.net.xml:42, 52.18 m), where lane ``highwayahead_0`` runs on at y = -1.60 (merge.net.xml:46).  A vehicle that SUMO
moves along these lanes reports exactly these (x, y) pairs through TraCI -- the positions the reference's planner and its trained
actors see."""
import numpy as np

RAMP_LANE_X = np.array([
    -250.47, -249.52, -248.53, -247.51, -246.45, -245.37, -244.27, -243.13, -241.95, -240.75, -239.53, -238.27, -237.00, -235.67, -234.34, -232.98,
    -231.59, -230.17, -228.73, -227.26, -225.76, -224.24, -222.71, -221.13, -219.55, -217.93, -216.29, -214.63, -212.95, -211.25, -209.52, -207.77,
    -206.01, -204.22, -202.41, -200.58, -198.74, -196.86, -194.98, -193.08, -191.15, -189.21, -187.26, -185.28, -183.29, -181.28, -179.26, -177.22,
    -175.16, -173.10, -171.02, -168.91, -166.81, -164.69, -162.55, -160.40, -158.23, -156.06, -153.87, -151.68, -149.47, -147.26, -145.02, -142.78,
    -140.54, -138.28, -136.02, -133.74, -131.47, -129.17, -126.88, -124.58, -122.27, -119.95, -117.63, -115.30, -112.97, -110.64, -108.30, -105.96,
    -103.61, -101.25, -98.90, -96.54, -94.18, -91.82, -89.46, -87.10, -84.74, -82.37, -80.00, -77.64, -75.28, -72.92, -70.56, -68.19, -65.85, -63.49,
    -61.14, -58.79, -56.45, -54.11, -51.77, -50.58,
])
RAMP_LANE_Y = np.array([
    28.47, 28.18, 27.88, 27.59, 27.30, 27.00, 26.71, 26.41, 26.12, 25.82, 25.53, 25.24, 24.94, 24.65, 24.36, 24.06, 23.76, 23.47, 23.18, 22.88,
    22.59, 22.30, 22.01, 21.71, 21.42, 21.12, 20.83, 20.54, 20.25, 19.96, 19.67, 19.38, 19.09, 18.80, 18.51, 18.23, 17.94, 17.65, 17.37, 17.08,
    16.80, 16.52, 16.23, 15.95, 15.67, 15.39, 15.11, 14.82, 14.55, 14.27, 13.99, 13.72, 13.45, 13.17, 12.90, 12.63, 12.36, 12.10, 11.83, 11.57,
    11.30, 11.04, 10.78, 10.52, 10.26, 10.01, 9.75, 9.50, 9.25, 8.99, 8.75, 8.50, 8.25, 8.01, 7.77, 7.53, 7.29, 7.05, 6.82, 6.58, 6.35, 6.12, 5.90,
    5.67, 5.45, 5.23, 5.01, 4.79, 4.58, 4.36, 4.15, 3.95, 3.74, 3.54, 3.33, 3.14, 2.94, 2.74, 2.55, 2.35, 2.17, 1.98, 1.80, 1.71,
])
JUNCTION_ENTRY = (-50.58, 1.71)       # end of ramp_0 = start of :mergenode_1_0
JUNCTION_EXIT = (1.50, -1.60)         # end of :mergenode_1_0 = start of highwayahead_0
HIGHWAY_LANE_Y = -1.60


def lane_polyline():
    """Ego route centre line from the ramp's start to the junction exit: (x[105], y[105], arc[105])."""
    x = np.append(RAMP_LANE_X, JUNCTION_EXIT[0])
    y = np.append(RAMP_LANE_Y, JUNCTION_EXIT[1])
    arc = np.concatenate([[0.0], np.cumsum(np.hypot(np.diff(x), np.diff(y)))])
    return x, y, arc


def point_at_arc(s):
    """(x, y) at arc length ``s`` from the ramp's start (beyond the junction exit: along y = -1.60)."""
    x, y, arc = lane_polyline()
    if s >= arc[-1]:
        return float(x[-1] + (s - arc[-1])), float(y[-1])
    return float(np.interp(s, arc, x)), float(np.interp(s, arc, y))
