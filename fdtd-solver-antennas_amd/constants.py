"""Physical constants (SI).  Same values the reference uses in antenna_sim/physics.py:9-12
and that openEMS exports as openEMS.physical_constants (C0, EPS0, MUE0)."""
import math

C0 = 299_792_458.0
MU0 = 4.0e-7 * math.pi
EPS0 = 1.0 / (MU0 * C0 * C0)
ETA0 = math.sqrt(MU0 / EPS0)
