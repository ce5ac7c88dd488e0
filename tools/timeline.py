#!/usr/bin/env python3
"""Per-wave timeline of ONE graph-replayed decode step (measurement build: `python voxtral-mini-realtime-rs_amd/build.py timeline`,
run with VOX_LIB=.../libvoxtral_hip_timeline.so, which this script selects itself).  Every q4_gemv / attn_decode launch of the step
stamps s_memrealtime (100 MHz, chip-global) per wave at: 0 kernel entry, 1 activation vector staged (GEMV) / scores done (attention),
2 first row group consumed (GEMV) / softmax done, 3 wave done.  Printed per kernel class, averaged over the 26 layers:
gap = first wave entry - last wave exit of the previous kernel; skew = last entry - first entry; x = median(stamp1 - own entry);
g1 = median(stamp2 - own entry); span = last exit - first entry (the kernel's in-flight time); all in microseconds."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("VOX_DECODE_UNROLL", "1")   # one step per graph: the slot bookkeeping below assumes it
os.environ.setdefault("VOX_LIB", os.path.join(ROOT, "voxtral-mini-realtime-rs_amd", "libvoxtral_hip_timeline.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
pkg = load_package(); L = pkg.lib(); S = pkg.synth
path = os.path.join(os.environ.get("VOX_BENCH_DIR", "/tmp"), "vox_bench_full_q4_seed42.gguf")
if not os.path.exists(path):
    S.write_synthetic_gguf(path + ".tmp", S.ModelDims(), seed=42); os.replace(path + ".tmp", path)
ctx = pkg.Context(0); model = pkg.Q4ModelLoader.from_file(path).load(ctx)
t = pkg.TimeEmbedding(3072).embed(6.0)
x = S.synth_audio(float(os.environ.get("VOX_TL_SECONDS", "16")), seed=1234); dx = ctx.upload(x)
NS, NW = 400, 8192
BATCH = int(os.environ.get("VOX_TL_BATCH", "0"))      # > 0: the batched decode step (vox_transcribe_batch of BATCH clips) instead of single-stream
# warm-up on a SHORT clip first (Ada scales, workspaces, a graph without slots); the 16 s clip then needs a larger audio buffer, so its
# first call re-captures the decode graph -- with timeline slots: slot 0 = prefill lm_head, 1..131 the eager step, 132..262 the CAPTURED step
xs = S.synth_audio(4.0, seed=99); dxs = ctx.upload(xs); model.transcribe_audio(None, t, device_ptr=dxs, n_samples=xs.size); ctx.free(dxs)
if BATCH:
    clips = [S.synth_audio(16.0, seed=1234 + i) for i in range(BATCH)]; ptrs = [ctx.upload(c) for c in clips]; lens = [c.size for c in clips]
    model.transcribe_batch(None, t, device_ptrs=ptrs, n_samples=lens)                      # warm-up (workspaces)
pkg._lib.check(L.vox_debug_timeline_start(ctx.h, NS, NW))
if BATCH:
    outs = model.transcribe_batch(None, t, device_ptrs=ptrs, n_samples=lens); ids = outs[0]      # slots: first lm_head, the eager step, then the CAPTURED step (rewritten by every replay)
else:
    for _ in range(3):
        ids = model.transcribe_audio(None, t, device_ptr=dx, n_samples=x.size)
buf = np.zeros((NS, NW, 4), dtype=np.uint64); used = C.c_int32(); meta = np.zeros((NS, 4), dtype=np.int32)
pkg._lib.check(L.vox_debug_timeline_fetch(ctx.h, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(used), meta.ctypes.data_as(C.c_void_p)))
tm = model.timings()
print(f"ids {len(ids)}, decode {tm['decode_ms']:.2f} ms -> {tm['decode_ms'] / max(len(ids), 1) * 1e3:.1f} us per step (instrumented build); slots used {used.value}")
TICK = 0.01   # us per s_memrealtime tick (100 MHz)
EPI = {0: "store", 1: "resid", 2: "swiglu", 3: "rope_kv", 4: "argmax", 5: "gelu", 6: "swiglu_xf", 7: "resid_xf", 8: "rope_kv+attn"}
# the captured 1-step graph = the slots stamped LAST (the final replay): order the live slots by first entry and keep the trailing run
# that fits in one step period
live = []
for k in range(used.value):
    b = buf[k].astype(np.int64); ok = b[:, 0] > 0
    if ok.any():
        live.append((int(b[ok, 0].min()), k))
live.sort()
t_end = live[-1][0]; period = tm['decode_ms'] / max(len(ids), 1) * 1e3 / TICK
step = [k for t0, k in live if t0 > t_end - 1.05 * period]
rows = {}; order = []; prev_end = None; t_first = None
for k in step:
    b = buf[k].astype(np.int64); ok = b[:, 0] > 0
    s0, s1, s2, s3 = (b[ok, i] for i in range(4))
    t0 = s0.min(); t_first = t0 if t_first is None else t_first
    name = ("attn" if meta[k][0] == 1 else f"{'skinny' if meta[k][0] == 2 else 'gemv'} {meta[k][2]}x{meta[k][3]} {EPI.get(int(meta[k][1]), meta[k][1])}")
    if name not in rows:
        rows[name] = []; order.append(name)
    rows[name].append(dict(gap=(t0 - prev_end) * TICK if prev_end is not None else np.nan, skew=(s0.max() - t0) * TICK,
                           x=float(np.median(s1 - s0)) * TICK, xmax=float((s1 - t0).max()) * TICK,
                           g1=float(np.median(s2[s2 > 0] - s0[s2 > 0])) * TICK if (s2 > 0).any() else np.nan,
                           end_med=float(np.median(s3 - t0)) * TICK, span=(s3.max() - t0) * TICK, wgs=int(ok.sum())))
    prev_end = s3.max()
print(f"last replayed step: {len(step)} launches, first entry -> last exit {(prev_end - t_first) * TICK:.1f} us")
print(f"{'kernel':34s} {'n':>3s} {'wgs':>5s} {'gap':>6s} {'skew':>6s} {'x_med':>6s} {'x_last':>7s} {'g1_med':>7s} {'end_med':>8s} {'span':>6s}")
tot = 0.0
for n in order:
    r = rows[n]
    m = {k: float(np.nanmean([e[k] for e in r])) for k in r[0]}
    tot += (np.nan_to_num(m["gap"]) + m["span"]) * len(r)
    print(f"{n:34s} {len(r):3d} {int(m['wgs']):5d} {m['gap']:6.2f} {m['skew']:6.2f} {m['x']:6.2f} {m['xmax']:7.2f} {m['g1']:7.2f} {m['end_med']:8.2f} {m['span']:6.2f}")
print(f"sum of (gap + span) over the step: {tot:.1f} us")
ctx.free(dx); model.close(); ctx.close()
