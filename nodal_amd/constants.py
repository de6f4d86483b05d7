"""Netlist grammar and device type codes.

Values follow the reference's CSV grammar (reference nodal/constants.py:7-38):
column positions, per-type field counts and the OPMODEL macro parameters.
The integer type codes (`TYPE_CODE`) are new: they are what the HIP stamping
kernel reads from the structure-of-arrays component table.
"""

# --- CSV column positions (reference nodal/constants.py:7-16) ---------------
NCOL, TCOL, VCOL, ACOL, BCOL, CCOL, DCOL, PCOL = range(8)

# --- component families (reference nodal/constants.py:18-22) ----------------
NODE_TYPES_CC = ["CCCS", "CCVS"]  # current-controlled: carry a driver name
NODE_TYPES_DEP = ["VCVS", "VCCS"] + NODE_TYPES_CC  # have control nodes
NODE_TYPES_ANOM = ["E"] + NODE_TYPES_DEP  # own a branch-current unknown
NODE_TYPES = ["A", "R"] + NODE_TYPES_ANOM + ["OPAMP", "OPMODEL"]

# exact number of CSV fields per type (reference nodal/constants.py:23-33)
NODE_ARGS_NUMBER = dict(
    R=5, A=5, E=5, VCCS=7, VCVS=7, CCCS=8, CCVS=8, OPAMP=7, OPMODEL=7
)

# --- OPMODEL macro (reference nodal/constants.py:36-38) ---------------------
OPMODEL_RI = 1e7  # input resistance, ohm
OPMODEL_RO = 10  # output resistance, ohm
OPMODEL_GAIN = 1e5  # open-loop gain

# --- device type codes (new; consumed by csrc/stamp.hip) --------------------
# VCCS shares VCVS's code on purpose: the reference dispatches VCCS rows to
# the VCVS stamp (reference nodal/nodal.py:377-378).
T_R, T_A, T_E, T_VCVS, T_CCVS, T_CCCS = range(6)
TYPE_CODE = {
    "R": T_R,
    "A": T_A,
    "E": T_E,
    "VCVS": T_VCVS,
    "VCCS": T_VCVS,
    "CCVS": T_CCVS,
    "CCCS": T_CCCS,
}
GROUND = -1  # node index meaning "this lead is the ground node"
