"""Equivalent resistance between two nodes, and the `nodal-resistance [-s] FILE`
command line entry (reference nodal/equiv.py:22-85)."""

import argparse
import os
import sys
from copy import deepcopy

import nodal_amd as n

parser = argparse.ArgumentParser(
    description="Calculate equivalent resistance using nodal analysis"
    "\n"
    "Label nodes as '1' and 'g' to mark where to connect to the network."
)
parser.add_argument("netlist_path", metavar="FILE", help="csv file describing the resistive network")
parser.add_argument("-s", "--sparse", action="store_true", help="use a sparse matrix")


# sparse sweeps of at least this many pairs over more than this many unknowns run on several device contexts
SWEEP_LANES = 1  # device contexts a long sparse sweep is spread over.  1 since round 4: the block iteration
                 # (csrc/sagg_multi.h, sixteen pairs per launch sequence on ONE hierarchy) beats three contexts with a
                 # hierarchy each -- 192 pairs on the 1e6-node grid: 0.54 s against 0.66 s, 40 pairs at 5.8e4 nodes: 22
                 # against ~70 ms.  NODAL_SWEEP_LANES=3 brings the lanes back (tests keep them exercised).
SWEEP_LANES_MIN_PAIRS = 12
SWEEP_LANES_MIN_UNKNOWNS = 50_000


def check_resistive(netlist):
    """True iff every component of the netlist is a resistor."""
    if getattr(netlist, "_fast", False):
        return bool((netlist._type.astype(str) == "R").all())
    return all(comp.type == "R" for comp in netlist.components.values())


def equivalent_resistance(netlist, a, b, sparse=False):
    """Resistance seen between nodes `a` and `b`: drive 1 A from b to a, solve,
    return e(a) - e(b) (reference nodal/equiv.py:31-61).

    Raises ValueError if the netlist holds anything but resistors and KeyError if
    a node is unknown.  As in the reference the probe source is named "a1", node
    numbering is not recomputed, and a node counts as grounded only if it is
    literally labelled "g"."""
    if not check_resistive(netlist):
        raise ValueError("Network is not resistive")
    for node in (a, b):
        if node not in netlist.nodenum and node != netlist.ground:
            raise KeyError(f"Node `{node}` not found in netlist")
    probed = deepcopy(netlist)
    probed.process_component(["a1", "A", "1", a, b])
    solution = n.Circuit(probed, sparse=sparse).solve()
    potential = [0, 0]
    for i, node in enumerate((a, b)):
        if node != "g":
            potential[i] = solution.result[solution.nodenum[node]]
    return potential[0] - potential[1]


def equivalent_resistance_sweep(netlist, pairs, sparse=False):
    """Equivalent resistance for many node pairs of one resistive network.

    Same results as `[equivalent_resistance(netlist, a, b, sparse) for a, b in pairs]`
    (and the same exceptions for a non-resistive network or an unknown node), but G is
    assembled once and factorised once (dense) / its multigrid hierarchy built once
    (sparse): only the 1 A probe source depends on the pair (SURVEY.md section 8f N1;
    the reference deep-copies the netlist and rebuilds everything per pair,
    nodal/equiv.py:50-53).  Returns a list of floats."""
    import numpy as np
    if not check_resistive(netlist):
        raise ValueError("Network is not resistive")
    ia, ib = [], []
    for a, b in pairs:
        for node in (a, b):
            if node not in netlist.nodenum and node != netlist.ground:
                raise KeyError(f"Node `{node}` not found in netlist")
        # the reference treats a node as grounded only if it is literally "g"
        ia.append(-1 if a == "g" else netlist.nodenum[a])
        ib.append(-1 if b == "g" else netlist.nodenum[b])
    circuit = n.Circuit(netlist, sparse=sparse)
    lanes = 1
    if sparse and len(ia) >= SWEEP_LANES_MIN_PAIRS and circuit._handle.n > SWEEP_LANES_MIN_UNKNOWNS:
        lanes = int(os.environ.get("NODAL_SWEEP_LANES", SWEEP_LANES))
    if lanes == 1:
        res, info = circuit._handle.solve_pairs(ia, ib, dense=not sparse)
    else:
        # A long sweep over a large network: every pair is one multigrid-preconditioned CG solve, whose coarse
        # levels leave most of the GPU idle.  Several device contexts (each with its own copy of the matrix and
        # of the hierarchy: a GB at 1e6 nodes, of 288), one host thread each, take the pairs in turn: three
        # solves in flight deliver 1.6 x the pairs per second of one (`concurrent`).
        import threading
        circuits = [circuit] + [n.Circuit._clone_of(circuit) for _ in range(lanes - 1)]
        parts = [list(range(k, len(ia), lanes)) for k in range(lanes)]
        out, infos, errors = [None] * lanes, [0] * lanes, []

        def work(k):
            try:
                sel = parts[k]
                out[k], infos[k] = circuits[k]._handle.solve_pairs([ia[i] for i in sel], [ib[i] for i in sel], dense=False)
            except BaseException as e:  # noqa: BLE001  (re-raised on the calling thread)
                errors.append(e)

        threads = [threading.Thread(target=work, args=(k,)) for k in range(lanes)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        res = np.empty(len(ia))
        for k in range(lanes):
            res[parts[k]] = out[k]
        info = max(infos)
    if info > 0 and not sparse:
        if not n.is_connected(netlist):
            raise n.UnconnectedCircuitError
        raise np.linalg.LinAlgError("Singular matrix")
    return [np.float64(r) for r in res]


def main(argv=None):
    args = parser.parse_args(argv)
    try:
        netlist = n.Netlist(args.netlist_path)
    except FileNotFoundError:
        sys.exit(1)
    try:
        r = equivalent_resistance(netlist, "1", "g", sparse=args.sparse)
    except ValueError:
        print("Invalid netlist\n")
        print("Resistors are the only component allowed in the circuit")
        sys.exit(1)
    except KeyError as e:
        print("Invalid netlist\n")
        print(e.args[0])
        sys.exit(1)
    print(f"R = {r}")


if __name__ == "__main__":
    main()
