"""Device memory across create / solve / destroy cycles of contexts on the three paths (development aid): the
figure must not grow from round to round."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from nodal_amd import _ffi, generators as gen
t = gen.grid_table(700)
t5 = gen.cfg5_table(300)
def used():
    free, total = torch.cuda.mem_get_info(0)
    return (total - free) / 2**20
torch.cuda.init()
base = used()
for r in range(12):
    h = _ffi.Handle(0); h.upload(t); h.run(False); h.synchronize(); h.close()
    h = _ffi.Handle(0); h.upload(t5); h.run(False); h.synchronize(); h.close()
    h = _ffi.Handle(0); h.upload(gen.grid_table(60)); h.run(True); h.synchronize(); h.close()
    if r in (0, 1, 5, 11): print(f"round {r}: device memory in use {used() - base:.0f} MiB above start", flush=True)
