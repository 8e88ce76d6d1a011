#!/usr/bin/env python3
"""Where a SMALL STARK proof's wall time goes that the kernels do not account for: wall seconds per C-ABI call (begin / second-round
columns / finish) against the HIP-event stage times inside each, SHA-256 table of 2^log_n rows.  usage: stark_host_gap.py [log_n]"""
import ctypes, json, sys, time
import numpy as np
sys.path.insert(0, ".")
import vectorx_amd as vx
from vectorx_amd import sha256_air, stark_chips

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 11
ctx = vx.Context(0)
msgs = [bytes([i]) * 55 for i in range(max(1, (1 << log_n) // 72 - 1))]
trace, pis, _ = sha256_air.generate_trace(log_n, msgs)
stark = sha256_air.make_stark(log_n)
tab = stark_chips.ResidentTable(ctx, stark, trace, pis, "sha256")
L, vp = vx.lib(), ctypes.c_void_p
for _ in range(3):
    tab.prove()
def one(prof):
    ctx.prof_enable(prof); ctx.prof_reset(); ctx.sync()
    t = [time.perf_counter()]
    sess = vp(); chal = np.zeros_like(tab.chal)
    assert L.vx_stark_begin(ctx._h, ctypes.cast(stark.desc_ptr, vp), vp(tab.d_trace), 1, tab.pis.ctypes.data if tab.pis.size else None, chal.ctypes.data, ctypes.byref(sess)) == 0
    t.append(time.perf_counter())
    st_begin = {k: v["ms"] for k, v in ctx.prof().items()} if prof else {}
    api = stark.run_aux_gpu(ctx, tab.d_trace, chal[:stark.desc.num_aux_challenges], tab.d_aux)
    t.append(time.perf_counter())
    st_aux = {k: v["ms"] for k, v in ctx.prof().items()} if prof else {}
    out = np.empty(tab.cap, dtype=np.uint8); nb = ctypes.c_size_t(tab.cap)
    assert L.vx_stark_finish2(sess, vp(tab.d_aux), 1, api.ctypes.data if api.size else None, None, out.ctypes.data, ctypes.byref(nb)) == 0
    t.append(time.perf_counter())
    st_all = {k: v["ms"] for k, v in ctx.prof().items()} if prof else {}
    L.vx_stark_session_free(sess)
    t.append(time.perf_counter())
    ctx.prof_enable(False)
    return [round((b - a) * 1e3, 3) for a, b in zip(t, t[1:])], (round(sum(st_begin.values()), 3), round(sum(st_aux.values()) - sum(st_begin.values()), 3),
                                                                 round(sum(st_all.values()) - sum(st_aux.values()), 3)), st_all
for prof in (False, True, False, True):
    rows = [one(prof) for _ in range(5)]
    walls = np.median(np.array([r[0] for r in rows]), axis=0)
    rec = {"log_n": log_n, "prof": prof, "wall_ms[begin, aux, finish, free]": [round(float(x), 3) for x in walls], "total": round(float(walls.sum()), 3)}
    if prof:
        rec["kernel_ms[begin, aux, finish]"] = [round(float(x), 3) for x in np.median(np.array([r[1] for r in rows]), axis=0)]
        rec["stages"] = {k: round(v, 3) for k, v in sorted(rows[-1][2].items(), key=lambda kv: -kv[1])}
    print(json.dumps(rec))
