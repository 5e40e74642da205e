"""Runner of scripts/probe/grid_phase_probe.hip: six dependent weight-streaming phases as six launches (one hipGraph) vs ONE
persistent launch with grid barriers between the phases, for three phase sizes.  python scripts/grid_phase_probe.py   (GPU box; builds the probe with hipcc)"""
import ctypes as C
import os
import subprocess
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = "/tmp/grid_phase_probe.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", os.path.join(R, "scripts/probe/grid_phase_probe.hip"), "-o", so])
lib = C.CDLL(so)
lib.grid_phase_launches.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
lib.grid_phase_persistent.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
nwg = torch.cuda.get_device_properties(dev).multi_processor_count
P, REPS = 6, 20
counters = torch.zeros(16, dtype=torch.int32, device=dev); status = torch.zeros(4, dtype=torch.int32, device=dev)
for kb in (8, 40, 128):                                                      # per workgroup and phase: 2 / 10 / 33 MB per phase on 256 CUs
    n4 = kb * 1024 // 16
    layers = 12                                                              # cycle 12 "layers" of weights so that nothing stays cache-resident
    w = torch.randn(layers, P, nwg, n4 * 4, device=dev)
    xa = torch.zeros(nwg * 16, device=dev); xb = torch.zeros(nwg * 16, device=dev)
    res = {}
    outs = {}
    for form in ("launches", "persistent"):
        def run(st, l):
            wp = w[l % layers].data_ptr()
            if form == "launches":
                rc = lib.grid_phase_launches(wp, n4, xa.data_ptr(), xb.data_ptr(), nwg, P, st)
            else:
                rc = lib.grid_phase_persistent(wp, n4, xa.data_ptr(), xb.data_ptr(), nwg, P, counters.data_ptr(), status.data_ptr(), st)
            assert rc == 0, rc
        xa.zero_(); xb.zero_()
        run(torch.cuda.current_stream().cuda_stream, 0)
        torch.cuda.synchronize()
        outs[form] = (xa.clone(), xb.clone())
        assert int(status[0]) == 0, hex(int(status[0]))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            st = torch.cuda.current_stream().cuda_stream
            for r in range(REPS):
                run(st, r)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / REPS)
        assert int(status[0]) == 0, hex(int(status[0]))
        res[form] = best
    same = all(torch.equal(a, b) for a, b in zip(outs["launches"], outs["persistent"]))
    mb = kb * 1024 * nwg / 1e6
    print(f"{P} phases x {mb:5.1f} MB ({kb:3d} KB per workgroup, {nwg} workgroups): six launches {res['launches']:6.2f} us   one persistent launch with 5 grid barriers "
          f"{res['persistent']:6.2f} us   ratio {res['persistent'] / res['launches']:.2f}   per seam {(res['persistent'] - res['launches']) / 5:+.2f} us   same results: {same}", flush=True)
