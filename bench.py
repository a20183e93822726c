#!/usr/bin/env python
"""Benchmark of the CSM generation hot path on MI355X -- BASELINE.json metric
"audio frames/sec (csm-1b, 512-frame ctx, greedy)".

A *step* = one frame-step of the hot path: the decoder loop (31 weight passes, as in the reference) that emits codebooks 0..31 of one
audio frame for every resident sequence, plus the backbone step that consumes that frame -- exactly one
iteration of the reference's `generate()` loop (modeling_csm.py:644-690), replayed from one hipGraph.
Workload at N=1: BASELINE configs[1] -- csm-1b, bf16 weights, B=1, 512-frame synthetic context
(128 text + 384 audio frames, seed 2), greedy (topk=1, T=1.0).  The 512-frame prefill happens before the
timed region (reported as `prefill_ms`); the context is resident in the KV cache when timing starts.

  python bench.py --gpus N --steps K --warmup W        (driver contract; N>1 via torch.distributed.run)

Multi-GPU = embarrassingly parallel batch split: every rank generates its own
utterance(s) with replicated weights, no data-path collective; RCCL only gathers the finished frames.

Prints ONE JSON line on rank 0.  `roofline` prices the graph launch (one frame-step) against HBM:
algorithmic bytes per step = bytes_step(B, L) of SURVEY.md section 8-d (weights once per pass + KV + the
embedding rows), L = mean cached length over the timed steps; duration from HIP events recorded on the
engine stream around the timed replays.  `cpu_baseline` = the oracle (a restatement of the reference's
CPU path, bit-exact against it on the golden vectors) timed on this host's cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from csm_hf_amd import CSMConfig, CSMModel  # noqa: E402
from csm_hf_amd.synth import synth_state_dict, synth_context  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling


def bytes_step(cfg: CSMConfig, B: int, L: float, wbytes: int = 2, kvbytes: int = 2, ebytes: int = 2) -> float:
    """Algorithmic bytes of one frame-step.
    The engine replaces the 31 per-codebook projection GEMVs by a table row read (4 KiB each), so the
    31 x projection term of the SURVEY formula is NOT counted; the position-0 projection is."""
    bc, dc = cfg.backbone_config, cfg.decoder_config
    C, V = cfg.audio_num_codebooks, cfg.audio_vocab_size

    def stack(lc):
        per = (lc.num_attention_heads + 2 * lc.num_key_value_heads) * lc.head_dim * lc.hidden_size \
            + lc.num_attention_heads * lc.head_dim * lc.hidden_size + 3 * lc.intermediate_size * lc.hidden_size \
            + 2 * lc.hidden_size * 1  # norms (counted at weight width, as SURVEY does)
        return lc.num_hidden_layers * per + lc.hidden_size
    w_bb = stack(bc) * wbytes
    w_dec = stack(dc) * wbytes
    w_heads = (V * bc.hidden_size + dc.hidden_size * bc.hidden_size + (C - 1) * V * dc.hidden_size) * wbytes
    w_step = w_bb + (C - 1) * w_dec + w_heads    # 31 decoder weight passes are algorithmically required (reference:
    # 1 two-token + 30 one-token forwards); the engine's v1 runs 32 one-token passes -- the extra pass is NOT counted
    kv_bb = bc.num_hidden_layers * 2 * bc.num_key_value_heads * bc.head_dim * kvbytes          # per position
    kv_dec = dc.num_hidden_layers * 2 * dc.num_key_value_heads * dc.head_dim * kvbytes
    dec_reads = kv_dec * sum(range(1, C + 1))
    emb = (C * bc.hidden_size * ebytes) + (C - 1) * dc.hidden_size * 4   # embedding rows stay bf16 with fp8 linears
    return w_step + B * (kv_bb * L + dec_reads + emb)


def cpu_baseline(cfg, model, ids, mask, frames: int, gpu_tokens, budget_s: float = 40.0, want_bf16: bool = True):
    """Oracle (checker + CPU baseline only) on the host cores, fp32 arithmetic on the same weights: decode
    frames/s after the same prefill.  Bounded: one prefill, a 1-frame probe per candidate thread count, then
    at most `frames` frames / `budget_s` seconds with the fastest thread count (M=1 GEMVs do not scale to 256
    threads; the count actually used is reported as `cores`)."""
    import copy
    from oracle import csm_oracle as O
    ncpu = os.cpu_count() or 1
    sd = {k: v.detach().to("cpu", torch.float32) for k, v in model.state_dict().items()}
    C = cfg.audio_num_codebooks

    def step(prev, cache):
        row = torch.cat([prev, torch.zeros(ids.shape[0], 1, dtype=torch.long)], 1).unsqueeze(1)
        m1 = torch.zeros(ids.shape[0], 1, C + 1, dtype=mask.dtype)
        m1[:, :, :C] = 1
        return O.generate_frame(sd, cfg, row, m1, 1.0, 1, cache, True)

    with torch.inference_mode():
        torch.set_num_threads(min(ncpu, 32))
        t0 = time.perf_counter()
        out = O.generate_frame(sd, cfg, ids, mask, 1.0, 1, None, True)
        t_prefill = time.perf_counter() - t0
        toks = [out.samples]
        probe = {}
        for nt in sorted({min(ncpu, n) for n in (8, 16, 32, 64)}):
            torch.set_num_threads(nt)
            c2 = copy.deepcopy(out.cache)
            t0 = time.perf_counter()
            step(toks[-1], c2)
            probe[nt] = time.perf_counter() - t0
        best_nt = min(probe, key=probe.get)
        torch.set_num_threads(best_nt)
        cache = out.cache
        t0 = time.perf_counter()
        done = 0
        while done < frames and time.perf_counter() - t0 < budget_s / 2:
            out = step(toks[-1], cache)
            cache = out.cache
            toks.append(out.samples)
            done += 1
        dt_s = time.perf_counter() - t0
    f32 = dict(value=round(done * ids.shape[0] / dt_s, 3), unit="frames/s", threads_used=best_nt, frames=done, arith="f32")
    if gpu_tokens is not None:
        n = min(len(toks), gpu_tokens.shape[1])
        f32["first_frames_equal_gpu"] = bool(torch.equal(torch.stack(toks[:n], 1), gpu_tokens[:, :n].cpu()))
    sample = (f"csm-1b, same {ids.shape[1]}-frame context, decode frames after prefill (prefill {t_prefill:.2f}s excluded), "
              f"B={ids.shape[0]}, greedy; thread-count probe " + ", ".join(f"{k}t:{1 / v:.2f}fps" for k, v in probe.items()))
    rec = dict(value=f32["value"], unit="frames/s", cores=best_nt, threads_used=best_nt, host_cores=ncpu,
               host_cpu_model=_cpu_model(), kind="port", arith="f32", sample=f"{done} frames; " + sample, f32=f32)
    # The reference's own dtype for this configuration is bf16 (README.md:73), and that is the faster CPU path: when the
    # model is bf16, `value` is the bf16 leg -- the same oracle in bf16 arithmetic, same thread count.
    # Its greedy stream is not comparable token for token (bf16 logits tie and the reference breaks ties with the RNG,
    # SURVEY.md section 8-c): the fp32 leg beside it carries the token comparison with the GPU.
    if want_bf16:
        try:
            sdb = {k: v.to(torch.bfloat16) for k, v in sd.items()}
            with torch.inference_mode():
                out = O.generate_frame(sdb, cfg, ids, mask, 1.0, 1, None, True)
                cache, prev = out.cache, out.samples
                t0 = time.perf_counter()
                done_b = 0
                while done_b < 2 * frames and time.perf_counter() - t0 < budget_s / 2:
                    row = torch.cat([prev, torch.zeros(ids.shape[0], 1, dtype=torch.long)], 1).unsqueeze(1)
                    m1 = torch.zeros(ids.shape[0], 1, C + 1, dtype=mask.dtype)
                    m1[:, :, :C] = 1
                    out = O.generate_frame(sdb, cfg, row, m1, 1.0, 1, cache, True)
                    cache, prev = out.cache, out.samples
                    done_b += 1
                rec["bf16"] = dict(value=round(done_b * ids.shape[0] / (time.perf_counter() - t0), 3), unit="frames/s",
                                   threads_used=best_nt, frames=done_b, arith="bf16")
            rec["value"], rec["arith"] = rec["bf16"]["value"], "bf16"
            rec["sample"] = f"{done_b} frames in bf16 arithmetic (the reference's dtype; fp32 leg of {done} frames beside it); " + sample
        except Exception as ex:   # never let the bf16 leg break the bench line: the fp32 figure stays in `value`
            rec["bf16"] = {"error": str(ex)[:200]}
    return rec


def run_config4(model, cfg, rank, world, dist, dev, ctx: int, frames: int):
    """BASELINE configs[3]: a [rows, ctx, 33] batch of utterances sharded over the ranks through `generate_sharded`
    (the real engine on every rank, RCCL only gathers the finished frames), greedy, stop disabled.  Two legs:
    weak = 16 rows per GPU (16 x world rows in total), strong = 128 rows in total.  Times the whole call (prefill +
    decode + gather, max over ranks) and, separately, the decode frame-steps from the engine's HIP events."""
    from csm_hf_amd.sharded import generate_sharded, shard_rows, MAX_ROWS_PER_PASS
    out = {}
    # exact = three exact bf16 planes per activation + fp32 KV cache (the mode the parity tests pin); exact_bf16kv = the same planes on
    # the DEFAULT cache of a bf16 checkpoint (bf16, the reference's own cache dtype: CSMModel.kv_dtype = "auto"); bf16 = decode_precision
    # AND prefill_precision "bf16" on the bf16 cache, the reference's own arithmetic class for the whole call.  Reported beside, never
    # instead of, the exact record.
    modes = [("exact", "exact", torch.float32)]
    if os.environ.get("CSM_BENCH_NO_DECODE_BF16") != "1":
        modes += [("exact_bf16kv", "exact", torch.bfloat16), ("bf16", "bf16", torch.bfloat16)]
    legs = (("weak", 16 * world), ("strong", 128))
    data = {leg: None for leg, _ in legs}
    for leg, rows in legs:
        ids, mask = synth_context(cfg, rows, ctx // 4, ctx - ctx // 4, seed=4)
        data[leg] = (ids.to(dev), mask.to(dev))
    for name, prec, kvd in modes:
        model.decode_precision = prec
        model.prefill_precision = prec
        model.kv_dtype = kvd
        for leg, rows in legs:
            ids, mask = data[leg]
            a0, a1 = shard_rows(rows, rank, world)
            walls = []
            for it in range(2):          # first pass sizes the engine and captures the graph (untimed)
                if dist is not None:
                    dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                toks = generate_sharded(model, ids, mask, max_new_frames=frames, temperature=1.0, topk=1,
                                        stop_on_all_zeros=False)
                torch.cuda.synchronize()
                if dist is not None:
                    dist.barrier()
                walls.append(time.perf_counter() - t0)
            dec_ms = model._engine.last_generate_ms()
            tm = torch.tensor([walls[-1], dec_ms / 1e3], dtype=torch.float64, device=dev)
            if dist is not None:
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            assert toks.shape == (rows, frames, cfg.audio_num_codebooks), toks.shape
            passes = max(1, -(-(a1 - a0) // MAX_ROWS_PER_PASS))
            rows_pass = min(a1 - a0, MAX_ROWS_PER_PASS)
            step_ms = float(tm[1]) * 1e3 / frames                 # one frame-step of the last engine pass (rows_pass rows)
            by = bytes_step(cfg, rows_pass, ctx + (frames - 1) / 2.0 + 1, kvbytes=4 if kvd == torch.float32 else 2)
            rec = {"rows_total": rows, "rows_per_gpu": a1 - a0, "frames": frames,
                   "ms_per_step_decode": round(step_ms, 4), "rows_per_engine_pass": rows_pass,
                   "roofline_frac_of_8TBs": round(by / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   "frames_per_s_end_to_end": round(rows * frames / float(tm[0]), 1),
                   "frames_per_s_decode_only": round(rows * frames / (float(tm[1]) * passes), 1),
                   "wall_s": round(float(tm[0]), 4), "decode_ms_last_pass": round(float(tm[1]) * 1e3, 2),
                   "engine_passes_per_gpu": passes, "kv_dtype": "f32" if kvd == torch.float32 else "bf16",
                   "tokens_checksum": int(toks.to(torch.int64).sum().item())}
            if name == "exact":
                out[leg] = rec
            else:
                key = "default_bf16_kv" if name == "exact_bf16kv" else "decode_precision_bf16"
                out[leg][key] = {k: rec[k] for k in ("ms_per_step_decode", "roofline_frac_of_8TBs", "frames_per_s_end_to_end",
                                                     "frames_per_s_decode_only", "wall_s", "kv_dtype", "tokens_checksum")}
                out[leg][key]["prefill_precision"] = prec
    model.decode_precision = "exact"
    model.prefill_precision = "exact"
    model.kv_dtype = torch.float32
    return out


def run_b1_default_kv(model, cfg, ids, mask, ctx: int, W: int, K: int, topk: int, temperature: float):
    """The headline workload (B = 1 per GPU, `ctx`-frame context, K timed frame-steps after W) in the SHIPPED DEFAULT of a bf16 checkpoint:
    bf16 KV cache (`CSMModel.kv_dtype = "auto"`), exact fp32 activations.  The headline itself stays on the fp32 cache because that mode
    reproduces the reference's token stream bit for bit (`parity.equal_all`); this record is what `from_pretrained(..., torch.bfloat16)`
    users get, priced against the bytes of ITS cache width.  Parity of this mode: tests/test_gpu_default_mode.py (tolerance protocol)."""
    B = ids.shape[0]
    model.kv_dtype = "auto"
    try:
        eng = model._ensure_engine(B, ctx + W + K + 2, W + K + 1, B * ctx)
        assert eng.kv_dtype == torch.bfloat16
        eng.reset()
        eng.set_kv_start([0] * B)
        eng.prefill(ids, mask, want_outputs=False)
        s = eng.sampling(temperature=temperature, topk=topk, seed=1234)
        eng.generate(s, W, True)
        eng.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.generate(s, K, True)
        eng.sync()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        hip_ms = eng.last_generate_ms()
        by = bytes_step(cfg, B, ctx + W + (K - 1) / 2.0 + 1, kvbytes=2)
        st = eng.prefetch_stats()
        toks = eng.read_frames(0, W + K)
        return {"kv_dtype": "bf16", "ms_per_step": round(wall / K * 1e3, 4), "hip_event_ms_per_step": round(hip_ms / K, 4),
                "frames_per_s": round(B * K / wall, 2), "algorithmic_bytes_per_step": int(by),
                "roofline_frac_of_8TBs": round(by / (hip_ms / K * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "weight_streamer": {"gave_up": st["gave_up"], "finished": st["finished"], "disabled": st["health"]["disabled"]},
                "tokens_checksum": int(toks.to(torch.int64).sum().item()),
                "parity": "tolerance protocol of SURVEY 8-c against the reference fixtures: tests/test_gpu_default_mode.py"}
    finally:
        model._drop_engine()
        model.kv_dtype = torch.float32


def run_configs_235(model, cfg, dev):
    """End-to-end records of BASELINE configs[1], [2] and [4] through the public API (`CSMModel.generate`: prefill + frame loop +
    the host-side bookkeeping of the call), each timed on its second call (the first sizes the engine and captures the graph).
    Bounded: a few seconds each.  (configs[3] is `config4`, run_config4.)"""
    import torch
    out = {}

    def timed(ids, mask, n, **kw):
        best = None
        for it in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            toks = model.generate(ids, mask, max_new_frames=n, stop_on_all_zeros=False, **kw)
            torch.cuda.synchronize()
            best = time.perf_counter() - t0
        dec_ms = model._engine.last_generate_ms()
        assert tuple(toks.shape) == (ids.shape[0], n, cfg.audio_num_codebooks), tuple(toks.shape)
        return best, dec_ms, int(toks.to(torch.int64).sum().item())

    try:
        # configs[1]: csm-1b bf16, 512-frame prefill + 200 frames greedy, B = 1 -- both prefill precisions
        ids, mask = synth_context(cfg, 1, 128, 384, seed=2)
        ids, mask = ids.to(dev), mask.to(dev)
        rec = {}
        for prec in ("exact", "bf16"):
            model.prefill_precision = prec
            wall, dec_ms, ck = timed(ids, mask, 200, temperature=1.0, topk=1)
            rec[f"prefill_{prec}"] = {"wall_s": round(wall, 4), "decode_ms": round(dec_ms, 2), "prefill_and_host_ms": round(wall * 1e3 - dec_ms, 2),
                                      "frames_per_s_end_to_end": round(200 / wall, 1), "tokens_checksum": ck}
        model.prefill_precision = "exact"
        rec["workload"] = "csm-1b bf16, B=1, 512-frame context prefilled INSIDE the timed call + 200 frames greedy, CSMModel.generate, fp32 KV (exact mode)"
        out["config2"] = rec
        # configs[2]: B = 16 voice-cloning style context (text + audio frames), top-k 50, T = 1.0, hipGraph outer frame step
        ids, mask = synth_context(cfg, 16, 128, 384, seed=3)
        ids, mask = ids.to(dev), mask.to(dev)
        rec = {}
        for name, prec, kvd in (("exact", "exact", torch.float32), ("default_bf16_kv", "exact", "auto"), ("decode_precision_bf16", "bf16", "auto")):
            model.decode_precision = prec
            model.prefill_precision = prec
            model.kv_dtype = kvd
            wall, dec_ms, ck = timed(ids, mask, 100, temperature=1.0, topk=50, seed=5)
            rec[name] = {"wall_s": round(wall, 4), "ms_per_step_decode": round(dec_ms / 100, 4), "frames_per_s_end_to_end": round(16 * 100 / wall, 1),
                         "frames_per_s_decode_only": round(16 * 100 / (dec_ms / 1e3), 1), "tokens_checksum": ck}
        model.decode_precision = "exact"
        model.prefill_precision = "exact"
        model.kv_dtype = torch.float32
        rec["workload"] = "csm-1b bf16, B=16, 512-frame text+audio context + 100 frames, top-k 50 / T 1.0 (device Philox), CSMModel.generate"
        out["config3"] = rec
        # configs[4]: fp8 (e4m3 + row scales) linear weights, 2048-frame prefill + 500 frames greedy, B = 1
        model.weight_format = "fp8"
        ids, mask = synth_context(cfg, 1, 512, 1536, seed=6)
        ids, mask = ids.to(dev), mask.to(dev)
        rec = {}
        for prec in ("exact", "mxfp8"):
            model.prefill_precision = prec
            wall, dec_ms, ck = timed(ids, mask, 500, temperature=1.0, topk=1)
            rec[f"prefill_{prec}"] = {"wall_s": round(wall, 4), "decode_ms": round(dec_ms, 2), "ms_per_step_decode": round(dec_ms / 500, 4),
                                      "prefill_and_host_ms": round(wall * 1e3 - dec_ms, 2), "frames_per_s_end_to_end": round(500 / wall, 1),
                                      "tokens_checksum": ck}
            by5 = bytes_step(cfg, 1, 2048 + (500 - 1) / 2.0 + 1, wbytes=1, kvbytes=4)
            rec[f"prefill_{prec}"]["roofline"] = {"bound": "hbm", "kernel": "frame-step hipGraph, fp8-e4m3 linear weights", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                                  "algorithmic_bytes_per_step": int(by5), "achieved": round(by5 / (dec_ms / 500 * 1e-3) / 1e9, 1),
                                                  "frac": round(by5 / (dec_ms / 500 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        rec["workload"] = "csm-1b fp8-e4m3 linear weights, B=1, 2048-frame context prefilled inside the timed call + 500 frames greedy, CSMModel.generate"
        out["config5"] = rec
    except Exception as ex:      # informative sub-records: never break the bench line
        out["error"] = f"{type(ex).__name__}: {str(ex)[:300]}"
    finally:
        model.prefill_precision = "exact"
        model.decode_precision = "exact"
        model.weight_format = "native"
        model.kv_dtype = torch.float32
    return out


def lib_sha256() -> str:
    """sha256 of the kernel SOURCES this library was built from (csm_hf_amd.build.sources_sha256): the PMC / timeline records name the
    build they were measured on by it -- a rebuild of the same sources is the same build, whatever its bytes are"""
    from csm_hf_amd.build import sources_sha256
    try:
        return sources_sha256()
    except OSError:
        return "unreadable"


def traffic_record(batch: int, ctx: int, weights: str, opts=()):
    """the committed PMC record (profiles/hbm_traffic.json) of this workload, or None.  A record carries the sha256 of the
    library it was measured on (`lib_sha256`, written by tools/pmc_record.py) and the engine options of its run; the caller
    marks the figure stale when the running build differs."""
    pmc = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        recs = json.load(open(pmc)).get("records", [])
    except Exception:
        return None
    for r in recs:
        if r.get("batch") == batch and r.get("ctx") == ctx and r.get("weights", "bf16") == weights and \
                sorted(r.get("opts", [])) == sorted(opts):
            return r
    return None


def launch_kinds(batch: int):
    """per-launch-kind table of the frame-step (gap / body / us per step by kernel and grid) from the committed in-step timeline
    of this batch size (profiles/launch_kinds_b<B>.json <- tools/b1_timeline.py --json), marked stale when measured on another build"""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", f"launch_kinds_b{batch}.json")))
    except Exception:
        return None
    rec["stale"] = rec.get("src_sha256") != lib_sha256()
    return rec


def attach_traffic(dst: dict, rec: dict, algorithmic: float = None):
    """`traffic` fields of a roofline / config4 record from a committed PMC record; `traffic_stale` = the record was
    measured on another build of the library (or names none)."""
    dst["traffic"] = rec["hbm_bytes_per_step"]
    if algorithmic:
        dst["traffic_over_algorithmic"] = round(rec["hbm_bytes_per_step"] / float(algorithmic), 3)
    dst["traffic_source"] = rec.get("source", "profiles/hbm_traffic.json")
    dst["traffic_src_sha256"] = rec.get("src_sha256")
    dst["traffic_commit"] = rec.get("commit")
    dst["traffic_stale"] = rec.get("src_sha256") != lib_sha256()


def pin_to_gpu_numa_node(local: int):
    """Keep this rank's host thread (one hipGraph replay loop per GPU, eight on a node) on the NUMA node of its GPU.
    Returns a short description for the record; never fails the run."""
    try:
        pr = torch.cuda.get_device_properties(local)
        bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return f"gpu {bdf}: no NUMA node reported, affinity unchanged"
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return f"gpu {bdf}: node {node} has no allowed CPU, affinity unchanged"
        os.sched_setaffinity(0, cpus)
        return f"gpu {bdf}: NUMA node {node}, {len(cpus)} CPUs"
    except Exception as ex:      # no sysfs entry / not permitted: run unpinned
        return f"unpinned ({type(ex).__name__})"


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


PREFILL_WARM, PREFILL_SAMPLES = 3, 5


def timed_prefill(eng, ids, mask, B):
    """(median, min) in ms of PREFILL_SAMPLES context prefills after PREFILL_WARM untimed ones.  Measured on the 512-frame
    context: after a precision switch the time drifts down by ~8 % over the first five calls (3.07 3.05 2.87 2.87 2.81 ms
    towards 2.75), with occasional +10 % outliers -- a single sample, or the first few, is not the steady value.
    Outside the timed region; informational."""
    ts = []
    for i in range(PREFILL_WARM + PREFILL_SAMPLES):
        eng.reset()
        eng.set_kv_start([0] * B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.prefill(ids, mask, want_outputs=False)
        eng.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("[bench] prefill samples (ms, first %d untimed): " % PREFILL_WARM + " ".join(f"{t:.2f}" for t in ts), file=sys.stderr, flush=True)
    ts = sorted(ts[PREFILL_WARM:])
    return ts[len(ts) // 2], ts[0]


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: re-run this file under torch.distributed.run with one rank per
    GPU.  Fails loudly (non-zero) when the box has fewer than N devices -- never reports a 1-GPU run as N."""
    import subprocess
    one_dev = os.environ.get("CSM_BENCH_ONE_DEVICE") == "1"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and not (one_dev and have >= 1):
        print(f"[bench] --gpus {n} requested but only {have} GPU(s) visible", file=sys.stderr, flush=True)
        return 3
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=1, help="sequences per GPU (BASELINE configs[1] = 1; configs[2,3] = 16)")
    ap.add_argument("--ctx", type=int, default=512)
    ap.add_argument("--topk", type=int, default=1)
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--kv-dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--weights", default="bf16", choices=["bf16", "fp8"], help="fp8 = e4m3fn linears + row scales (config 5)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lean", action="store_true", help="profiling runs: no cpu_baseline leg and no dominant-kernel chain "
                    "(its launches would be counted with the step's own kernels)")
    ap.add_argument("--cpu-frames", type=int, default=8)
    ap.add_argument("--config4", type=int, default=-1, help="1/0: also run the BASELINE configs[3] sub-record (batch of "
                    "utterances through generate_sharded, 16 rows/GPU weak + 128 rows strong); default: on for the "
                    "default workload (B = 1, 512-frame context) at every N, off with --lean")
    ap.add_argument("--config4-frames", type=int, default=100)
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        print(f"[bench] --gpus {a.gpus} does not match WORLD_SIZE {world}", file=sys.stderr, flush=True)
        sys.exit(3)
    dist = None
    dist_backend = None
    if "WORLD_SIZE" in os.environ:      # started by a launcher (torch.distributed.run), also with ONE rank: the same
        # process-group code path (RCCL init, MAX all-reduce, all-gather) runs at every world size
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # test hook (1-GPU box): CSM_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and uses gloo, which exercises the
        # rank slicing / barrier / MAX-reduce / gather logic of this file without a second GPU; never set by the driver
        if os.environ.get("CSM_BENCH_ONE_DEVICE") == "1":
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
            dist_backend = "gloo"
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
            dist_backend = "nccl"
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)      # before the first allocation of this process: everything below lives on `dev`
    affinity = pin_to_gpu_numa_node(local)

    cfg = CSMConfig()
    B, K, W = a.batch, a.steps, a.warmup
    n_text = a.ctx // 4
    ids_all, mask_all = synth_context(cfg, world * B, n_text, a.ctx - n_text, seed=2)
    ids, mask = ids_all[rank * B:(rank + 1) * B], mask_all[rank * B:(rank + 1) * B]

    t0 = time.perf_counter()
    sd = synth_state_dict(cfg, seed=0, dtype=torch.bfloat16, device=dev, bf16_representable=True)
    model = CSMModel(cfg)
    model.load_state_dict(sd)
    del sd
    model.kv_dtype = torch.float32 if a.kv_dtype == "f32" else torch.bfloat16
    model.weight_format = "fp8" if a.weights == "fp8" else "native"
    eng = model._ensure_engine(B, a.ctx + W + K + 2, W + K + 1, B * a.ctx)
    for o in a.opt:
        k, v = o.split("=")
        eng.set_option(k, int(v))
    t_setup = time.perf_counter() - t0
    print(f"[bench] rank {rank}: setup {t_setup:.1f}s", file=sys.stderr, flush=True)

    eng.reset()
    eng.set_kv_start([0] * B)
    eng.prefill(ids[:, :min(a.ctx, 128)], mask[:, :min(a.ctx, 128)], want_outputs=False)   # cold start: code objects load here
    # context prefill, timed in both precisions (bf16 weights): "bf16" = activations rounded to bf16 at the GEMM inputs
    # (one MFMA pass), "exact" = fp32 activations as three bf16 planes.  The benchmarked run continues from the EXACT one.
    prefill_ms_bf16 = prefill_min_bf16 = None
    if a.weights != "fp32":
        eng.set_option("prefill_bf16", 1)
        prefill_ms_bf16, prefill_min_bf16 = timed_prefill(eng, ids, mask, B)
        eng.set_option("prefill_bf16", 0)
    prefill_ms_mx = prefill_min_mx = None
    if a.weights == "fp8":
        # BASELINE configs[4] "fp8 weights (CDNA4 fp8 MFMA)": the context GEMMs on v_mfma_scale_f32_16x16x128_f8f6f4 with
        # weights AND activations in OCP MX-fp8 (csrc/gemm_mx.h) -- its own accuracy class, opt-in; timed beside the others
        eng.enable_mx(model.state_dict())
        eng.set_option("prefill_bf16", 1)
        eng.set_option("prefill_mx", 1)
        prefill_ms_mx, prefill_min_mx = timed_prefill(eng, ids, mask, B)
        eng.set_option("prefill_mx", 0)
        eng.set_option("prefill_bf16", 0)
    prefill_ms, prefill_min = timed_prefill(eng, ids, mask, B)   # the benchmarked run continues from this (exact) context
    s = eng.sampling(temperature=a.temperature, topk=a.topk, seed=1234)
    use_graph = not a.no_graph
    eng.generate(s, W, use_graph)
    eng.sync()

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.generate(s, K, use_graph)
    eng.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    wall = time.perf_counter() - t0
    hip_ms = eng.last_generate_ms()
    print(f"[bench] rank {rank}: prefill {prefill_ms:.1f} ms, {K} steps wall {wall:.3f}s hip {hip_ms:.1f} ms", file=sys.stderr, flush=True)

    # host-side replay overhead of THIS rank: wall time of the K replays minus the HIP-event time of the same replays
    host_over = (wall - hip_ms / 1e3) / K * 1e3
    tm = torch.tensor([wall, hip_ms / 1e3, host_over], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    wall_max, hip_max, host_over_max = float(tm[0]), float(tm[1]), float(tm[2])
    from csm_hf_amd.engine import live_engines
    assert live_engines() == 1, f"rank {rank}: {live_engines()} engines alive in the timed process (one engine per process and GPU)"

    pf_stats = eng.prefetch_stats()
    pf_stats["disabled_reason"] = None if pf_stats["health"]["disabled"] == 0 else pf_stats["health"]["reason"]
    # per-rank step times: when a multi-GPU node runs this, the line shows that the process group saw N ranks
    # and which rank was the slowest
    per_rank_ms = [wall / K * 1e3]
    ranks_seen = 1
    if dist is not None:
        ranks_seen = dist.get_world_size()
        mine = torch.tensor([wall / K * 1e3], dtype=torch.float64, device=dev)
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        per_rank_ms = [float(g[0]) for g in got]
    toks = eng.read_frames(0, W + K)
    all_toks = toks
    if dist is not None:      # the only RCCL traffic: gather finished frames, off the timed path
        gathered = [torch.empty_like(toks) for _ in range(world)]
        dist.all_gather(gathered, toks)
        all_toks = torch.cat(gathered, 0)
        assert all_toks.shape == (world * B, W + K, cfg.audio_num_codebooks)
    # position-weighted checksum of every rank's frames (tests compare it with solo runs of the same rows)
    wgt = torch.arange(1, B * (W + K) * cfg.audio_num_codebooks + 1, device=dev, dtype=torch.int64).reshape(B, W + K, -1)
    checks = [int((all_toks[r * B:(r + 1) * B] * wgt).sum()) for r in range(world)]

    if rank == 0:
        L_mean = a.ctx + W + (K - 1) / 2.0 + 1          # positions read by the backbone step of timed frame i
        kvb = 4 if a.kv_dtype == "f32" else 2
        by = bytes_step(cfg, B, L_mean, wbytes=1 if a.weights == "fp8" else 2, kvbytes=kvb)   # KV priced at its stored width
        by_survey = bytes_step(cfg, B, L_mean, wbytes=1 if a.weights == "fp8" else 2, kvbytes=2)
        step_s = hip_max / K
        achieved = by / step_s / 1e9
        out = {
            "metric": "audio frames/sec (csm-1b, 512-frame ctx, greedy)",
            "value": round(world * B * K / wall_max, 3),
            "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(wall_max / K * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("fp8-e4m3 linear weights (per-row scales), bf16 embeddings" if a.weights == "fp8" else "bf16 weights")
                     + ", f32 activations/accumulate, " + a.kv_dtype + " KV"
                     + ("; the timed decode steps are exact in the fp8 weights; the opt-in MX-fp8 context prefill (prefill_ms_mxfp8) is "
                        "pinned per operation (quantiser bit-exact, GEMM <= 5e-5 sum|a||b| vs fp64) and end to end by last_h rel-L2 "
                        "<= 1.25 x the oracle+MX simulation's own self-distance (0.20) and <= 0.30 (DESIGN section 2)"
                        if a.weights == "fp8" else ""), "data": "synthetic",
            "config": {"workload": f"csm-1b ({a.weights}), B={B}/GPU, {a.ctx}-frame synthetic context prefilled (untimed), "
                                   f"{K} timed frame-steps after {W} warm-up, topk={a.topk} T={a.temperature}, "
                                   f"hipGraph={'on' if use_graph else 'off'}, kv_dtype={a.kv_dtype}, decode_precision=exact, "
                                   f"engine options: {' '.join(sorted(a.opt)) if a.opt else 'defaults'}",
                       "batch_per_gpu": B, "context_frames": a.ctx, "parallelism": f"batch-split x{world}"},
            "tokens_checksum_per_rank": checks,
            "dist_backend": dist_backend,      # null: single process without a launcher (no process group)
            "ranks_seen": ranks_seen, "ms_per_step_per_rank": [round(x, 4) for x in per_rank_ms],
            "slowest_rank": int(max(range(len(per_rank_ms)), key=lambda i: per_rank_ms[i])),
            "weight_streamer": pf_stats,
            "prefill_ms": round(prefill_ms, 2),
            "prefill_ms_statistic": f"median of {PREFILL_SAMPLES} calls after {PREFILL_WARM} untimed",
            "prefill_ms_min": round(prefill_min, 2),
            "prefill_ms_bf16_activations_min": None if prefill_min_bf16 is None else round(prefill_min_bf16, 2),
            "prefill_ms_bf16_activations": None if prefill_ms_bf16 is None else round(prefill_ms_bf16, 2),
            "prefill_ms_mxfp8": None if prefill_ms_mx is None else round(prefill_ms_mx, 2),
            "prefill_ms_mxfp8_min": None if prefill_min_mx is None else round(prefill_min_mx, 2),
            "hip_event_ms_per_step": round(step_s * 1e3, 4),
            "host_overhead_ms_per_step_max_over_ranks": round(host_over_max, 4),   # wall - HIP events: graph-replay jitter of the slowest host thread
            "host_affinity_rank0": affinity,
            "multi_gpu_note": ("one process per GPU, rows sharded, no data-path collective (DESIGN section 6); "
                               + ("this line is a 1-GPU measurement -- no N > 1 number has been measured on hardware by the builder"
                                  if world == 1 else f"{world} ranks over {dist_backend}")),
            "setup_s": round(t_setup, 1),
            "roofline": {"bound": "hbm", "kernel": "frame-step hipGraph (decoder loop + backbone step)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_of_measured_6290": round(achieved / 6290.0, 4),
                         "algorithmic_bytes_per_step": int(by), "algorithmic_bytes_per_step_bf16_kv": int(by_survey),
                         "traffic": None},
        }
        # the dominant kernel of the step by time: the decoder gate/up GEMV (124 launches per step, 33.55 MB each).
        # Its launch time is measured here with HIP events on the engine stream as a dependent chain of 200
        # launches in a hipGraph cycling over 512 MiB of weights (so it includes the launch boundary, like the step).
        if B == 1 and a.weights == "bf16" and not a.lean:
            try:
                dc = cfg.decoder_config
                us, wb = eng.bench_gemv(2 * dc.intermediate_size, dc.hidden_size, M=1, norm=True, epi=2)
                out["roofline"]["dominant_kernel"] = {
                    "name": "gemv1_kernel<bf16, NORM, SWIGLU> (decoder gate/up, 124 launches/step)",
                    "algorithmic_bytes_per_launch": wb, "us_per_launch_chain": round(us, 3),
                    "achieved": round(wb / us / 1e3, 1), "unit": "GB/s", "frac": round(wb / us / 1e3 / HBM_PEAK_GBS, 4),
                    "share_of_step_time": round(124 * us / (step_s * 1e6), 3),
                    "how": "side chain of 200 dependent launches of this kernel (csm_bench_gemv: no streamer beside it), NOT read from the "
                           "step; in the step itself (roofline.launch_kinds, profiles/r05_b1_timeline.md) the same launch costs 1.6 us gap + "
                           "3.8-3.9 us body with the streamer delivering part of its weights ahead (5.3 us of body without)"}
            except Exception as ex:      # never let the side measurement break the bench line
                out["roofline"]["dominant_kernel"] = {"error": str(ex)[:200]}
        # `traffic`: HBM bytes per frame-step from the PMC counters.  Counters cannot be read inside this process, so the
        # figure is the separately collected rocprofv3 --pmc measurement of THIS command (tools/collect_pmc.sh: FETCH_SIZE
        # and WRITE_SIZE in their own passes, FETCH_SIZE x 2 for gfx950's wide reads, decode kernels only), committed as
        # profiles/hbm_traffic.json; null when no record matches the workload of this run.
        rec = traffic_record(B, a.ctx, a.weights, a.opt)
        if rec is not None:
            attach_traffic(out["roofline"], rec, by)
        out["src_sha256"] = lib_sha256()
        lk = launch_kinds(B) if (a.ctx == 512 and a.weights == "bf16" and a.topk == 1) else None
        if lk is not None:
            out["roofline"]["launch_kinds"] = lk
        # parity of the benchmarked run against the reference's golden vectors (same context at rank 0, B=1)
        gpath = os.path.join(ROOT, "tests", "golden", "csm1b_cfg2_bf16w_fp32.npz")
        if B == 1 and a.ctx == 512 and a.topk == 1 and a.weights == "bf16" and os.path.exists(gpath):
            import numpy as np
            g = np.load(gpath)
            n = min(W + K, g["tokens"].shape[1])
            margin = (g["top_vals"][..., 0] - g["top_vals"][..., 1])[:n, 0].reshape(-1)
            ref = g["tokens"][0, :n].reshape(-1)
            mine = toks[0, :n].cpu().numpy().reshape(-1)
            low = np.nonzero(margin < 1e-4)[0]
            stop = int(low[0]) if len(low) else len(ref)
            out["parity"] = {"vs": "reference golden tokens (fp32 arithmetic, same weights)",
                             "samples_compared": stop, "equal": bool((mine[:stop] == ref[:stop]).all()),
                             "equal_all": bool((mine == ref).all())}
        if world == 1 and not a.no_cpu_baseline and not a.lean:
            out["cpu_baseline"] = cpu_baseline(cfg, model, ids, mask, a.cpu_frames, toks)
    # BASELINE configs[3] (batch of utterances sharded over the GPUs): every rank takes part
    want4 = a.config4 != 0 and B == 1 and a.ctx == 512 or a.config4 == 1   # default: on (the B = 16 per-GPU shape is the
    # one an 8-GPU curve multiplies -- it belongs in every driver-timed line); --config4 0 switches it off
    c4 = None
    b1_default = None
    if want4 and not a.lean and a.weights == "bf16" and a.kv_dtype == "f32" and use_graph:
        del eng
        eng = None
        b1_default = run_b1_default_kv(model, cfg, ids, mask, a.ctx, W, K, a.topk, a.temperature)
    if want4 and not a.lean and a.weights == "bf16":
        del eng
        c4 = run_config4(model, cfg, rank, world, dist, dev, a.ctx, a.config4_frames)
    e2e = None
    if want4 and not a.lean and a.weights == "bf16" and world == 1 and os.environ.get("CSM_BENCH_NO_E2E") != "1":
        e2e = run_configs_235(model, cfg, dev)
    if rank == 0:
        if b1_default is not None:
            out["default_bf16_kv"] = b1_default
        if e2e is not None:
            out.update(e2e)
        if c4 is not None:
            rec16 = traffic_record(16, a.ctx, "bf16")      # PMC record of the per-GPU shape of the weak leg (--batch 16)
            if rec16 is not None and c4["weak"]["rows_per_gpu"] == 16:
                attach_traffic(c4["weak"], rec16)
            lk16 = launch_kinds(16)
            if lk16 is not None and c4["weak"]["rows_per_gpu"] == 16:
                c4["weak"]["launch_kinds"] = lk16
            out["config4"] = c4
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
