"""`nodal.constants` of the reference (nodal/constants.py:7-38): the same values, from nodal_amd."""
from nodal_amd.constants import *  # noqa: F401,F403
