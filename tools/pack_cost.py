#!/usr/bin/env python3
"""development aid: what e2s_attach_reads costs with and without the reads' bit planes (E2S_TAIL_PACKED=0 under E2S_DEV=1), c3's 3 GB of reads
   tools/pack_cost.py [reads=20000000] [len=150]"""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from ebwt2snp_amd import capi
    R, L = int(sys.argv[2]), int(sys.argv[3])
    t = torch.randint(0, 4, (R, L), dtype=torch.uint8, device="cuda")
    t = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device="cuda")[t.long()] if R * L < 1 << 28 else (t * 0 + 65)
    torch.cuda.synchronize()
    h = capi.Handle(0)
    ts = []
    for i in range(4):
        t0 = time.perf_counter()
        h.attach_reads(t.data_ptr(), None, None, R, L, keep=(t,))
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("E2S_TAIL_PACKED=%s: e2s_attach_reads of %.2f GB: %s ms (first call allocates)" % (os.environ.get("E2S_TAIL_PACKED", "1"), R * L / 1e9, ", ".join("%.3f" % x for x in ts)))
else:
    R = sys.argv[1] if len(sys.argv) > 1 else "20000000"; L = sys.argv[2] if len(sys.argv) > 2 else "150"
    for sw in ("1", "0"):
        subprocess.run([sys.executable, __file__, "child", R, L], env=dict(os.environ, E2S_DEV="1", E2S_TAIL_PACKED=sw))
