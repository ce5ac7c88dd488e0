#!/usr/bin/env python3
"""Rewrites the round-6 number paragraphs of README.md / DESIGN.md from profiles/r06_bench_n1.json + r06_bench_kernel_stats.csv (so that the prose cites the committed line)."""
import csv, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_n1.json")))
b = d["batch"]; r = d["roofline"]; f = d["fleurs_like"]; fc = d["fleurs_like_cli"]; fs = d["fleurs_like_sessions"]; se = d["streaming_encoder"]
row = [x for x in csv.DictReader(open(os.path.join(ROOT, "profiles", "r06_bench_kernel_stats.csv"))) if "decode_engine_kernel" in x["Name"]][0]
ravg = float(row["AverageNs"]) / 1e3; rcalls = int(row["Calls"])
br = b["roofline"]
p = os.path.join(ROOT, "DESIGN.md"); s = open(p).read()
a = s.index("**Round-6 numbers** (1 × MI355X"); e = s.index("**Round-5 numbers** (1 × MI355X")
new = f'''**Round-6 numbers** (1 × MI355X, `profiles/r06_bench_n1.json`, `r06_bench_kernel_stats.csv`, `r06_batch16_kernel_stats.csv`, `r06_share81_kernel_stats.csv`; box-to-box spread ≈ ± 2 %): single clip
**RTF {d['rtf']:.4f}, {d['value']:.0f} tok/s** end-to-end (engine launch {r['avg_launch_us']:.1f} µs by HIP events — one event pair per position quarter around a graph replay of 50 launches, median of three passes of 200 —
{ravg:.1f} µs rocprofv3 average over {rcalls} launches of a second, profiled run on the same box = {r['achieved']/1000:.2f} TB/s = **{r['frac']:.2f} of 8 TB/s**, unchanged: no structural attempt was made on the single-stream engine this round — §7);
piecewise C loop {d['piecewise']['tok_per_s']:.0f} tok/s; **batch 16: {b['tok_per_s']:.0f} tok/s**, {b['ms_per_batch']:.1f} ms per batch (encoder {b['stage_ms']['encode_ms']:.1f} ms, §3.2; decode step {b['decode_step_ms']:.3f} ms); the `batch.roofline` block prices the WHOLE
batched step on algorithmic bytes incl. K / V: {br['algorithmic_bytes_per_step']/1e9:.2f} GB in {b['decode_step_ms']:.3f} ms = {br['achieved']/1000:.2f} TB/s = **{br['frac']:.2f}** (counter traffic {br['traffic']/1e9:.2f} GB: {br['traffic']/br['algorithmic_bytes_per_step']:.2f} ×); f32 path {d['f32']['tok_per_s']:.0f} tok/s.
**FLEURS-like corpus, ONE call, one session: {f['tok_per_s']:.0f} tok/s, {f['wall_s']:.2f} s for 647 clips** (un-chunked pipeline; front-end {f['last_call_stage_ms']['preprocess_ms']:.0f} + encoder {f['last_call_stage_ms']['encode_ms']:.0f} + prefill and {f['last_call_decode_steps']} decode steps {f['last_call_stage_ms']['decode_ms']:.0f} ms;
9 667 tok/s a round ago; this round: 10.4 k with the four-group wide step on 64 slots → 11.0 k with that step as two two-group chains → 12.3–12.4 k on 128 slots, two four-group chains per step, §3.3f); **on the reference
CLI's pipeline (1200-frame chunks as units, §3.3g) {fc['tok_per_s']:.0f} tok/s, {fc['wall_s']:.2f} s for 846 units = {fc['tok_per_s_vs_unchunked']:.2f} of the un-chunked figure; as TWO sessions of 64 slots on the GPU (`vox_model_set_sessions`, §3.3h,
`fleurs_like_sessions`): {fs['tok_per_s']:.0f} tok/s, {fs['wall_s']:.2f} s = × {fs['tok_per_s_vs_one_session']:.2f}, same ids** (two sessions and two chains are the same overlap: × 1.16–1.20 over the 64-slot single session, next to nothing on top of
128 slots) — the one-session runs carry their own `simulated_world`: a rank's 81-clip share {f['simulated_world']['predicted_wall_s']:.3f} s (un-chunked) / {fc['simulated_world']['predicted_wall_s']:.3f} s (CLI) → predicted 8-GPU scaling **{f['simulated_world']['predicted_scaling']:.2f} × / {fc['simulated_world']['predicted_scaling']:.2f} ×** of the
one-GPU run — lower than the 6.85–7.45 × of the round's earlier lines ONLY because the one-GPU run got 16 % faster: the share itself (two-group engine steps bound by its longest clip; too small for more groups
or a second session) is where it was.  Streaming encoder (§8 f2): {se['chunk_100_frames']['latency_ms_mean']:.1f} ms per 1 s chunk, {se['chunk_1200_frames']['latency_ms_mean']:.1f} ms per 12 s chunk, steady state with eviction.  PMC passes of both batched engine forms:
`profiles/r06_pmc_batch_engines.txt` (one group per launch: 2.17 GB of fabric traffic against 2.01 GB algorithmic = 1.07 ×; two groups: 3.98 GB against 2.42 GB = 1.65 × — the second pass over the packets leaves
the XCD L2s; whether the Infinity Cache serves it is not observable: `TCC_EA0_RDREQ_DRAM` counts requests to the DRAM address space, not cache misses).  `cpu_baseline`: {d['cpu_baseline']['value']:.2f} tok/s (128 host threads, 16 s clip).

'''
s = s[:a] + new + s[e:]; open(p, "w").write(s)
p = os.path.join(ROOT, "README.md"); s = open(p).read()
a = s.index("| BASELINE config | result | dominant kernel: achieved / peak |"); e = s.index("Round 6, what changed:")
tbl = f'''| BASELINE config | result | dominant kernel: achieved / peak |
|---|---|---|
| [1] single 16 s clip, f32 SafeTensors path | RTF {d['f32']['rtf']:.4f} · {d['f32']['tok_per_s']:.0f} tok/s | decode step 0.52 of the HBM peak (unchanged) |
| [2] single 16 s clip, Q4_0 (the `metric`) | RTF **{d['rtf']:.4f}** · **{d['value']:.0f} tok/s** end-to-end · piecewise C loop {d['piecewise']['tok_per_s']:.0f} tok/s · encode {d['stage_ms']['encode_ms']:.2f} ms, prefill 2.6 ms, decode step {r['decode_step_measured_ms']:.2f} ms | `decode_engine_kernel` {r['avg_launch_us']:.1f} µs per launch (HIP events around graph replays, median of 3 passes of 200; rocprofv3 {ravg:.1f} µs over {rcalls} launches of a profiled run on the same box) = {r['achieved']/1000:.2f} TB/s = **{r['frac']:.2f}** of the HBM peak — unchanged for the fourth round |
| [3] 16 × 16 s clips, Q4_0 | **{b['tok_per_s']:.0f} tok/s** · {b['ms_per_batch']:.1f} ms per batch (encode {b['stage_ms']['encode_ms']:.1f} + decode {b['stage_ms']['decode_ms']:.1f}) · decode step {b['decode_step_ms']:.3f} ms | whole batched step on ALGORITHMIC bytes incl. K / V (`batch.roofline`): {br['algorithmic_bytes_per_step']/1e9:.2f} GB in {b['decode_step_ms']:.3f} ms = {br['achieved']/1000:.2f} TB/s = **{br['frac']:.2f}**; counter traffic {br['traffic']/1e9:.2f} GB = {br['traffic']/br['algorithmic_bytes_per_step']:.2f} × |
| [4] 647 FLEURS-like clips, one GPU, ONE call — un-chunked pipeline | RTF {f['rtf']:.5f} · **{f['tok_per_s']:.0f} tok/s** · {f['wall_s']:.2f} s (round 5: 9 667) · rank share {f['simulated_world']['predicted_wall_s']:.3f} s → predicted 8-GPU scaling {f['simulated_world']['predicted_scaling']:.2f} × of the (now faster) one-GPU run | up to 128 decode slots; a step = one Q4 GEMM per operator for the groups of a chain, two chains of four groups on two streams: 4.70 ms per 128 slots (64 slots: 3.55 → 2.92 ms) |
| [4] the same corpus as **two concurrent sessions of 64 slots** (`vox_model_set_sessions(m, 2)`: hidden context + model replica + library thread; `DESIGN.md` §3.3h) | RTF {fs['rtf']:.5f} · **{fs['tok_per_s']:.0f} tok/s** · {fs['wall_s']:.2f} s = × {fs['tok_per_s_vs_one_session']:.2f} of the one-session call, same ids | the same overlap as two chains: × 1.16–1.20 over a 64-slot session, next to nothing on top of 128 slots |
| [4] the same corpus on the **reference CLI's pipeline** (file normalised once, 1200-frame chunks as units: 846 units) | RTF {fc['rtf']:.5f} · **{fc['tok_per_s']:.0f} tok/s** · {fc['wall_s']:.2f} s = {fc['tok_per_s_vs_unchunked']:.2f} of the un-chunked figure · rank share {fc['simulated_world']['predicted_wall_s']:.3f} s → {fc['simulated_world']['predicted_scaling']:.2f} × | — |
| streaming encoder (§8 f2), 120 s stream | {se['chunk_100_frames']['latency_ms_mean']:.1f} ms per 1 s chunk (RTF {se['chunk_100_frames']['rtf']:.4f}) · {se['chunk_1200_frames']['latency_ms_mean']:.1f} ms per 12 s chunk | — |
| CPU oracle on the box's 128 host cores (`cpu_baseline`) | {d['cpu_baseline']['value']:.2f} tok/s · {d['cpu_baseline']['total_s']:.0f} s per clip | |

'''
s = s[:a] + tbl + s[e:]; open(p, "w").write(s)
print("README.md / DESIGN.md refreshed from", d["value"], "tok/s;", f["tok_per_s"], "corpus tok/s")
