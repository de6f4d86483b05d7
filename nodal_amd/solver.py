"""Command line entry `nodal-solver [-s] FILE` (reference nodal/solver.py:7-31)."""

import argparse
import sys

import nodal_amd as n

parser = argparse.ArgumentParser(description="Solve electrical circuits using nodal analysis")
parser.add_argument("netlist_path", metavar="FILE", help="csv file describing the netlist")
parser.add_argument("-s", "--sparse", action="store_true", help="use a sparse matrix")


def main(argv=None):
    args = parser.parse_args(argv)
    try:
        netlist = n.Netlist(args.netlist_path)
    except FileNotFoundError:
        sys.exit(1)
    circuit = n.Circuit(netlist, sparse=args.sparse)
    try:
        solution = circuit.solve()
    except n.UnconnectedCircuitError:
        sys.exit(1)
    print(solution)


if __name__ == "__main__":
    main()
