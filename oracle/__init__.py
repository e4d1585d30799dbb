"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's hot path, used as the parity checker by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under uav_airvision_amd/
imports this package.
"""
