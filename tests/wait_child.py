"""Child process of tests/test_gpu_kernels.py::test_a_host_wait_that_runs_out_says_where_and_what.

Run with NODAL_WAIT_TIMEOUT_S set to a few microseconds (csrc/wait.hip reads it once per process): the first
host wait of a large solve that is not over at once runs into the bound.  The library must return NODAL_E_HIP
with the wait site and the last kernel enqueued on the stream in nodal_last_error, refuse further calls on
the handle, and nodal_destroy must not free under the running kernels.  Prints "wait child ok"."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from nodal_amd import _ffi, generators as gen  # noqa: E402


def main():
    assert float(os.environ["NODAL_WAIT_TIMEOUT_S"]) < 1e-3
    h = _ffi.Handle(0)
    table = gen.grid_table(700)
    msg = None
    try:
        h.upload(table)
        h.assemble_symbolic()
        h.assemble_numeric()
        h.solve_sparse()
    except _ffi.NodalHipError as e:
        assert e.status == _ffi.E_HIP, e
        msg = str(e)
    assert msg is not None, "no wait ran into a bound of a few microseconds"
    assert "timed out after" in msg and ".hip:" in msg, msg
    assert "last kernel enqueued on it: " in msg and "launch #" in msg, msg
    print("diagnostic:", msg)
    # the handle is unusable, and says so at once
    try:
        h.assemble_symbolic()
        raise AssertionError("a hung handle took another call")
    except _ffi.NodalHipError as e:
        assert e.status == _ffi.E_HIP and "timed out" in str(e), e
    time.sleep(0.5)  # (the device work was fine: let it drain before the process ends)
    print("wait child ok")


if __name__ == "__main__":
    main()
