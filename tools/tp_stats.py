#!/usr/bin/env python3
"""tp_stats.py -- rounds / walk lengths / cycles of the two-pass tracebacks, from a -DAT_TP_STATS=1 build (AT_LIB_PATH):
    AT_LIB_PATH=aligntools/c_amd/exp/libaligntools_hip_st.so python3 tools/tp_stats.py C2 C3 C4
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import aligntools.c_amd as A
from aligntools.c_amd.synth import workload_codes_torch, pack2_torch
sys.path.insert(0, ROOT)
import bench

dev = torch.device("cuda", 0)
for w in sys.argv[1:] or ["C2", "C3", "C4"]:
    mode, l1, l2, pairs, scoring, use_jump, sites, seed = bench.WORKLOADS[w]
    al = A.Aligner(0)
    al.set_scoring(*scoring, use_jump, sites)
    w1, w2 = (l1 + 15) // 16 + 1, (l2 + 15) // 16 + 1
    d_words = torch.zeros(pairs * (w1 + w2) + 4, dtype=torch.int32, device=dev)
    wv = d_words[:pairs * (w1 + w2)].view(pairs, w1 + w2)
    c = workload_codes_torch(mode, use_jump, seed, pairs, l1, l2, 0, dev)
    wv[:, :w1] = pack2_torch(c[:, :l1]); wv[:, w1:] = pack2_torch(c[:, l1:])
    d_woff1 = torch.arange(pairs, dtype=torch.int64, device=dev) * (w1 + w2)
    d_woff2 = d_woff1 + w1
    d_len1 = torch.full((pairs,), l1, dtype=torch.int32, device=dev)
    d_len2 = torch.full((pairs,), l2, dtype=torch.int32, device=dev)
    d_res = torch.zeros((5, pairs), dtype=torch.int32, device=dev)
    d_ops = torch.zeros(pairs * (l1 + l2) + 64, dtype=torch.uint8, device=dev)
    d_ops_off = torch.arange(pairs, dtype=torch.int64, device=dev) * (l1 + l2)
    stream = torch.cuda.current_stream().cuda_stream
    def run():
        al.align_batch_device(A.MODES[mode], pairs, d_words.data_ptr(), 2, d_woff1.data_ptr(), d_len1.data_ptr(), d_woff2.data_ptr(), d_len2.data_ptr(),
                              l1, l2, True, True, d_res[0].data_ptr(), d_res[1].data_ptr(), d_res[2].data_ptr(), d_res[3].data_ptr(),
                              d_ops.data_ptr(), d_ops_off.data_ptr(), d_res[4].data_ptr(), stream)
    run(); torch.cuda.synchronize()
    lib = al._lib
    out = (C.c_ulonglong * 8)()
    lib.at_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    lib.at_debug_counters(al._h, out)
    before = list(out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    lib.at_debug_counters(al._h, out)
    d = [int(out[k]) - before[k] for k in range(8)]
    items = max(1, d[1])
    print("%s: %.3f ms  %s" % (w, e0.elapsed_time(e1), al.last_config[:110]))
    if "walk kernel" in al.last_config:
        wv = max(1, d[1])
        if os.environ.get("TP_STATS_LEVEL") == "2":   # a -DAT_TP_STATS=2 build: the replays apart
            rd = max(1, d[2])
            print("   walk kernel: wavefronts %d  rounds/wavefront %.2f  kcycles/wavefront %.0f; per round: first loads + staging %.1f, set-up %.1f, step loop %.1f, walks %.1f"
                  % (d[1], d[2] / wv, d[3] / wv / 1e3, d[4] / rd / 1e3, d[5] / rd / 1e3, d[6] / rd / 1e3, d[7] / rd / 1e3))
        else:
            print("   walk kernel: wavefronts %d  rounds/wavefront %.2f  kcycles/wavefront %.0f (start-up %.0f, replays %.0f, walks %.0f), longest %.0f"
                  % (d[1], d[2] / wv, d[3] / wv / 1e3, d[4] / wv / 1e3, d[5] / wv / 1e3, d[6] / wv / 1e3, int(out[7]) / 1e3))
        al.close()
        continue
    print("   items %d  rounds/item %.2f  kcycles/item: forward %.0f  pass 2 %.0f (replay %.0f, walks %.0f of which block copies %.0f)"
          % (d[1], d[2] / items, d[7] / items / 1e3, d[5] / items / 1e3, d[6] / items / 1e3, d[3] / items / 1e3, d[4] / items / 1e3))
    al.close()
