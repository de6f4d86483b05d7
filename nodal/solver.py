"""`nodal.solver` of the reference (nodal/solver.py): the `nodal-solver` entry point."""
from nodal_amd.solver import main, parser  # noqa: F401

if __name__ == "__main__":
    main()
