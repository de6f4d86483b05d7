"""`nodal.equiv` of the reference (nodal/equiv.py): equivalent resistance and the
`nodal-resistance` entry point."""
from nodal_amd.equiv import check_resistive, equivalent_resistance, main, parser  # noqa: F401

if __name__ == "__main__":
    main()
