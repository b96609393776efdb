#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

Metric (BASELINE.json): "MNT4-753 G1 MSM scalar-muls/sec + 2^n NTT ms at 1/2/4/8 MI355X".
  value  = whole-job MSM throughput: every rank runs one complete MNT4-753 G1 MSM of 2^log_n
           (base, scalar) pairs per step on its own GPU (bases and scalars already resident in
           HBM), then the partial sums are all-gathered and folded (weak scaling, one exchange).
  ntt    = (extra object) 2^ntt_log_n-point MNT4-753 Fr NTT, device resident, ms per transform.
One "step" = one pass of the MSM hot path over one batch of synthetic input.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

def kernels_sha():
    """sha256 over the device sources: stamps profiles/*_pmc_traffic.json so that stale counters are not reported"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ginger-lib_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 measured)
FPMUL_PEAK_PER_S = 23.4e9       # measured 753-bit Montgomery products/s, profiles/r01_microbench_valu_rates.txt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log-n", type=int, default=20, help="log2 of MSM pairs per GPU")
    ap.add_argument("--ntt-log-n", type=int, default=24)
    ap.add_argument("--window", type=int, default=0, help="MSM window override (0 = auto)")
    ap.add_argument("--no-precompute", action="store_true",
                    help="skip gh_bases_precompute: time the per-window path (no shift table for the resident key)")
    ap.add_argument("--no-pipeline", action="store_true", help="issue the steps one by one instead of as one pipelined batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--no-2p24", action="store_true", help="skip the extra 2^24-pair object (BASELINE config 3's second size)")
    ap.add_argument("--no-g2", action="store_true", help="skip the extra G2 objects (MNT4-753 G2 2^20, MNT6-753 G2 2^19)")
    ap.add_argument("--curve", default="mnt4753_g1", choices=["mnt4753_g1", "mnt4753_g2", "mnt6753_g1", "mnt6753_g2"])
    ap.add_argument("--total-log-n", type=int, default=0,
                    help="strong scaling (BASELINE config 4): 2^total-log-n pairs in all, 2^total-log-n / N per GPU; overrides --log-n")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started as a plain `python bench.py --gpus N`: become the launcher.  One rank per GPU is started as a child
        # process tree (torch.distributed.run) BEFORE this process has touched the GPU; nothing is exec'ed over a
        # process that has.
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s)\n" % (args.gpus, world))
        sys.exit(2)
    dist = None
    cdist = None
    # The contract is ONE JSON line on stdout, but gloo ("[Gloo] Rank r is connected ...") and RCCL (its version banner)
    # write to file descriptor 1 from native code: point fd 1 at stderr for the whole run and keep the real stdout aside.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # Rendezvous, barrier and the max over ranks go over gloo (CPU); the data path -- the one exchange of partial sums --
    # is the library's own RCCL communicator behind the C ABI (include/ginger_hip_dist.h), so torch's RCCL is never loaded.
    # GH_DIST_BACKEND=gloo rehearses the N > 1 path on a one-GPU box (ranks share the card, the exchange goes over gloo).
    backend = os.environ.get("GH_DIST_BACKEND", "rccl")
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
        dist.barrier()

    import pyref
    from __graft_entry__ import _load_pkg
    gl = _load_pkg()
    import importlib
    distmod = importlib.import_module("ginger_lib_amd.dist")
    # raises if the HIP library or a gfx950 device is missing: no fallback.  (gloo rehearsal on a one-GPU
    # box: let the library map LOCAL_RANK modulo the device count.)
    share_card = backend != "rccl"
    if world > 1 and not share_card:
        import torch
        share_card = torch.cuda.device_count() < world      # fewer cards than ranks (a rehearsal): RCCL will refuse, see below
    gl.init(None if (world > 1 and share_card) else local_rank)
    exchange_note = None
    if world > 1:
        def _callback_transport():
            return distmod.CDist(gl, rank, world, transport="callback", allgather=distmod.gloo_allgather_bytes(dist))
        if backend == "rccl":
            from torch.distributed.distributed_c10d import _get_default_store
            import torch
            # bring the communicator up and push one exchange of the identity through it; every rank then agrees (over gloo)
            # whether RCCL is usable.  If any rank saw an error the whole job drops to the callback transport, and says so.
            err = ""
            try:
                cdist = distmod.CDist(gl, rank, world, transport="rccl", store=_get_default_store())
                ident = np.zeros(36 * pyref.CURVES[args.curve].deg, dtype=np.uint64)
                ident[12 * pyref.CURVES[args.curve].deg] = 1          # (0 : 1 : 0); any limbs do for the probe
                cdist.allgather_fold(args.curve, ident)
            except Exception as e:      # noqa: reported below
                err = "%s: %s" % (type(e).__name__, e)
            flag = torch.tensor([0 if err else 1])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                sys.stderr.write("bench.py rank %d: RCCL exchange unavailable (%s); using the gloo callback transport\n" % (rank, err or "another rank failed"))
                if cdist is not None:
                    cdist.shutdown()
                cdist = _callback_transport()
                exchange_note = "gloo callback (RCCL failed: %s)" % (err or "on another rank")
        else:
            cdist = _callback_transport()
    if args.window:
        gl.msm_set_window(args.window)

    curve = args.curve
    C = pyref.CURVES[curve]
    strong = args.total_log_n > 0
    if strong:
        if (1 << args.total_log_n) % world:
            sys.stderr.write("bench.py: 2^%d pairs do not split evenly over %d ranks\n" % (args.total_log_n, world))
            sys.exit(2)
        n = (1 << args.total_log_n) // world
    else:
        n = 1 << args.log_n
    # ---- synthetic inputs: n DISTINCT bases P_0 + i H along an addition chain, generated on the device
    #      (gh_bases_generate_chain: every G1 point is a valid base, multiples of the G2 generator stay in the subgroup);
    #      scalars uniform in [0, r) with the reference's sampling shape, different per rank.
    import support as S     # helpers only (layout conversion); the oracle is used in cpu_baseline alone

    def chain_key(crv, count, seed):
        Cc = pyref.CURVES[crv]
        prng = pyref.Rng(seed)
        xy, _ = S.bases_array(Cc, [Cc.mul(prng.next_u64() | 1, Cc.G), Cc.mul(prng.next_u64() | 1, Cc.G)])
        return gl.ResidentBases.chain(crv, xy[0], xy[1], count)

    rb = chain_key(curve, n, 1 + rank)
    scalars = S.random_scalars_np(n, seed=1000 + rank, below=C.order)
    ds = gl.DeviceBuffer(n * 96).upload(scalars)
    # The bases are a proving key: uploaded once, outside the timed region, and -- like the
    # layout conversion at upload -- expanded once into the shift table 2^(c w) P_i (gh_bases_precompute).
    # The per-window path (no table) is timed as well and reported under "per_window_path".
    plain = None
    table_info = None
    if not args.no_precompute and not args.window:
        for _ in range(max(1, args.warmup)):
            rb.msm_dev(ds, n)
        gl.load_library().gh_dev_sync()
        t0 = time.perf_counter()
        for _ in range(2):
            plain_out = rb.msm_dev(ds, n)
        dt = (time.perf_counter() - t0) / 2
        ptm = gl.msm_last_timing()
        plain = {"value": n / dt, "unit": "scalar-muls/s per GPU", "ms_per_step": dt * 1e3, "window_bits": ptm["window_bits"],
                 "num_windows": ptm["num_windows"], "accumulate_ms": ptm["accumulate_ms"], "steps": 2}
        t0 = time.perf_counter()
        c_tab = rb.precompute(0)
        rows = 752 // c_tab + 1
        table_info = {"window_bits": c_tab, "rows": rows, "bytes": rows * n * 208 * C.deg, "build_s": time.perf_counter() - t0,
                      "note": "one-time per resident key, outside the timed region (like the base upload)"}
        a1 = gl.proj_to_affine(curve, rb.msm_dev(ds, n))
        t0 = time.perf_counter()
        rb.msm_dev(ds, n)
        rb.msm_dev(ds, n)
        table_info["single_msm_ms"] = (time.perf_counter() - t0) / 2 * 1e3    # latency of one MSM, nothing to overlap with
        plain["note"] = "no shift table, MSMs issued one by one"
        a0 = gl.proj_to_affine(curve, plain_out)
        plain["same_affine_result_as_table_path"] = bool(a0[1] == a1[1] and (a0[0] == a1[0]).all())

    def proj_add(acc, p):
        return gl.proj_add(curve, acc, p)

    def run_steps(k):
        """k steps = k complete MSMs over the resident inputs.  They are handed to the library as ONE batch
        (gh_msm_resident_dev_batch, as a prover hands over its sequence of MSMs): the bucket sort of step
        i+1 and the bucket reduction + fold of step i-1 overlap the accumulation of step i on HIP streams.
        --no-pipeline issues them one by one.  Multi-GPU: one all-gather + fold of the partial sums per step."""
        if args.no_pipeline:
            partials, tms = [], []
            for _ in range(k):
                partials.append(rb.msm_dev(ds, n))
                tms.append(gl.msm_last_timing())
        else:
            partials = gl.msm_batch_dev([(rb, ds, n)] * k)
            tms = [gl.msm_batch_timing(i) for i in range(k)]
        totals = [cdist.allgather_fold(curve, p) if world > 1 else p for p in partials]
        if world > 1:
            exch_us.append(cdist.last_exchange_us)
        return totals, tms

    exch_us = []

    def sync():
        gl.load_library().gh_dev_sync()      # the library's streams (this process holds no other GPU work)
        if world > 1:
            dist.barrier()

    # which accumulation the library's automatic policy picked for this curve (include/ginger_hip.h: gh_msm_set_affine)
    bucket_mode = "affine rounds (aff_kernels.h) + projective finish" if (C.deg > 1 and os.environ.get("GH_AFFINE", "2") != "0") or os.environ.get("GH_AFFINE") == "1" \
        else "projective mixed additions"
    if args.warmup:
        run_steps(args.warmup)
    sync()
    t0 = time.perf_counter()
    results, tms = run_steps(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    result, tm = results[-1], tms[-1]
    acc_ms = [t["accumulate_ms"] for t in tms]
    phases = {"sort_ms": 0.0, "accumulate_ms": 0.0, "heavy_ms": 0.0, "reduce_ms": 0.0, "fold_ms": 0.0}
    for t in tms:
        for k in phases:
            phases[k] += t[k] / args.steps
    if world > 1:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = world * n * args.steps / elapsed
    tm_last = tm
    acc_avg_ms = float(np.mean(acc_ms))
    # bytes per pair = affine base + 96-byte scalar, read once: G1 288 B, MNT4 G2 480 B, MNT6 G2 672 B
    alg_bytes = (192.0 * C.deg + 96.0) * n
    achieved = alg_bytes / (acc_avg_ms * 1e-3) / 1e9
    madds = tm_last["accumulate_madds"]
    # base-field products per addition: projective mixed addition 11 (madd-1998-cmo); affine rounds 5 M + 1 S (+ the shared
    # inversion, not counted); tower fields: Fq2 product = 4 Fp products in the lane-pair form, Fq3 = 9
    per_add = (6 if bucket_mode.startswith("affine") else 11) * {1: 1, 2: 4, 3: 9}[C.deg]
    fpmul_rate = madds * per_add / (acc_avg_ms * 1e-3)

    # HBM traffic per launch of the dominant kernels: PMC counters collected with rocprofv3 in separate
    # passes on this same command (profiles/r01_pmc_traffic.json); null if that file is absent or the
    # workload differs from the profiled one (2^20 pairs / 2^24 points).
    traffic_acc = traffic_ntt = None
    traffic_src = "profiles/r02_pmc_traffic.json"
    try:
        tr = json.load(open(os.path.join(ROOT, traffic_src)))
        if tr.get("kernels_sha256") == kernels_sha():          # counters go stale when the kernels change: then null
            ent = tr.get("%s_2p%d" % (curve, n.bit_length() - 1))
            if ent and ent["window_bits"] == tm_last["window_bits"] and ent["bucket_sums"] == bucket_mode:
                traffic_acc = ent["fetch_bytes_per_msm"] + ent["write_bytes_per_msm"]
            ent = tr.get("ntt_2p%d" % args.ntt_log_n)
            if ent:
                traffic_ntt = ent["fetch_bytes_per_transform"] + ent["write_bytes_per_transform"]
    except Exception:
        pass

    out = {
        "metric": "MNT4-753 G1 MSM scalar-muls/sec + 2^n NTT ms at 1/2/4/8 MI355X",
        "value": value,
        "unit": "scalar-muls/s",
        "n_gpus": cdist.world_seen if cdist is not None else 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "u32 (29-bit limbs of 753-bit Montgomery residues, 64-bit accumulators)",
        "data": "synthetic",
        "config": {"workload": "%s VariableBaseMSM, %s (base,scalar) pairs per GPU%s, bases+scalars resident in HBM" % (
                       {"mnt4753_g1": "MNT4-753 G1", "mnt4753_g2": "MNT4-753 G2", "mnt6753_g1": "MNT6-753 G1", "mnt6753_g2": "MNT6-753 G2"}[curve],
                       "2^%d" % (n.bit_length() - 1) if n & (n - 1) == 0 else str(n),
                       " (2^%d in all, strong scaling)" % args.total_log_n if strong else ""),
                   "curve": curve, "pairs_per_gpu": n, "window_bits": tm_last["window_bits"], "num_windows": tm_last["num_windows"],
                   "resident_key_shift_table": table_info, "distinct_bases": n, "bucket_sums": bucket_mode, "parallelism": "pairs sharded by rank, 1 all-gather of partial sums" if world > 1 else "single GPU"},
        "roofline": {"kernel": "msm_accumulate_kernel (bucket accumulation of the %s MSM)" % curve, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_acc,
                     "traffic_note": "FETCH_SIZE + WRITE_SIZE bytes of the accumulation launches of one MSM from %s (separate rocprofv3 --pmc passes), null when that file was measured on other kernel sources (sha256 of ginger-lib_amd/csrc) or another workload" % traffic_src,
                     "avg_launch_ms": acc_avg_ms, "algorithmic_bytes_per_launch": alg_bytes,
                     "note": "integer-VALU bound by construction (SURVEY 8d): see valu"},
        "valu": {"achieved_fpmul_per_s": fpmul_rate, "peak_fpmul_per_s": FPMUL_PEAK_PER_S, "frac": fpmul_rate / FPMUL_PEAK_PER_S,
                 "fp_products_per_addition": per_add,
                 "note": "peak = measured rr29 Montgomery-product microbenchmark (profiles/r01_microbench_valu_rates.txt)"},
        "phases_ms": phases,
        "phases_note": "per-MSM device phases by HIP events on their streams; with the pipelined batch the phases of neighbouring "
                       "steps overlap (sort and reduce run beside the previous / next accumulation), so they do not add up to ms_per_step",
        "pipelined": not args.no_pipeline,
    }
    if world > 1:
        out["exchange"] = {"transport": exchange_note or ("rccl all-gather behind gh_partials_allgather_fold" if backend == "rccl" else "gloo (rehearsal)"),
                           "ranks_seen_by_transport": cdist.world_seen, "bytes_per_rank": 288 * C.deg,
                           "avg_us": float(np.mean(exch_us[-args.steps:])) if exch_us else None}
    if plain is not None:
        out["per_window_path"] = plain
        if not plain["same_affine_result_as_table_path"]:
            out["error"] = "table path and per-window path disagree"

    # ---- BASELINE config 3's second size: 2^24 pairs (per-window path and shift table c = 23, 115 GB), same line
    if not args.no_2p24 and rank == 0 and world == 1 and curve == "mnt4753_g1" and not args.window and args.log_n < 24:
        try:
            n24 = 1 << 24
            rb24 = chain_key(curve, n24, 77)
            s24 = S.random_scalars_np(n24, seed=2024, below=C.order)
            d24 = gl.DeviceBuffer(n24 * 96).upload(s24)
            del s24
            rb24.msm_dev(d24, n24)
            t1 = time.perf_counter()
            p_out = rb24.msm_dev(d24, n24)
            pw_ms = (time.perf_counter() - t1) * 1e3
            ptm = gl.msm_last_timing()
            t1 = time.perf_counter()
            c24 = rb24.precompute(0)
            build_s = time.perf_counter() - t1
            rb24.msm_dev(d24, n24)
            t1 = time.perf_counter()
            t_out = rb24.msm_dev(d24, n24)
            tb_ms = (time.perf_counter() - t1) * 1e3
            ttm = gl.msm_last_timing()
            t1 = time.perf_counter()
            gl.msm_batch_dev([(rb24, d24, n24)] * 3)
            bt_ms = (time.perf_counter() - t1) * 1e3 / 3
            a0, a1 = gl.proj_to_affine(curve, p_out), gl.proj_to_affine(curve, t_out)
            out["msm_2p24"] = {
                "workload": "MNT4-753 G1 VariableBaseMSM, 2^24 distinct pairs, resident",
                "per_window_path": {"value": n24 / pw_ms * 1e3, "ms": pw_ms, "window_bits": ptm["window_bits"], "num_windows": ptm["num_windows"],
                                    "accumulate_ms": ptm["accumulate_ms"]},
                "shift_table": {"value": n24 / tb_ms * 1e3, "ms": tb_ms, "window_bits": c24, "rows": 752 // c24 + 1,
                                "bytes": (752 // c24 + 1) * n24 * 208, "build_s": build_s, "accumulate_ms": ttm["accumulate_ms"],
                                "pipelined_batch_of_3": {"value": n24 / bt_ms * 1e3, "ms_per_msm": bt_ms}},
                "unit": "scalar-muls/s", "same_affine_result": bool(a0[1] == a1[1] and (a0[0] == a1[0]).all())}
            if not out["msm_2p24"]["same_affine_result"]:
                out["error"] = "2^24: table path and per-window path disagree"
            d24.free()
            rb24.free()
            gl.dev_trim()
        except gl.GingerHipError as e:          # e.g. not enough HBM next to another tenant: the object is absent, the line stands
            out["msm_2p24"] = {"error": str(e)}

    # ---- G2 (the prover's b_g2 MSM: BASELINE configs 4 / 5 shapes), one MSM at a time and as a pipelined batch of three
    if not args.no_g2 and rank == 0 and world == 1 and curve == "mnt4753_g1" and not args.window:
        out["g2"] = {}
        for crv, lg in (("mnt4753_g2", 20), ("mnt6753_g2", 19)):
            try:
                ng = 1 << lg
                Cg = pyref.CURVES[crv]
                rbg = chain_key(crv, ng, 31)
                dg = gl.DeviceBuffer(ng * 96).upload(S.random_scalars_np(ng, seed=4242, below=Cg.order))
                t1 = time.perf_counter()
                cg = rbg.precompute(0)
                build_s = time.perf_counter() - t1
                rbg.msm_dev(dg, ng)
                t1 = time.perf_counter()
                rbg.msm_dev(dg, ng)
                rbg.msm_dev(dg, ng)
                one_ms = (time.perf_counter() - t1) / 2 * 1e3
                gtm = gl.msm_last_timing()
                t1 = time.perf_counter()
                gl.msm_batch_dev([(rbg, dg, ng)] * 3)
                bt_ms = (time.perf_counter() - t1) * 1e3 / 3
                out["g2"][crv] = {"workload": "%s VariableBaseMSM, 2^%d pairs, resident key with shift table" % (crv, lg),
                                  "value": ng / bt_ms * 1e3, "unit": "scalar-muls/s", "ms_per_msm_pipelined": bt_ms, "single_msm_ms": one_ms,
                                  "window_bits": cg, "table_build_s": build_s, "bucket_sums": "affine rounds (aff_kernels.h) + projective finish" if os.environ.get("GH_AFFINE", "2") != "0" else "projective mixed additions",
                                  "phases_ms": {k: gtm[k] for k in ("sort_ms", "accumulate_ms", "reduce_ms", "fold_ms") if k in gtm}}
                dg.free()
                rbg.free()
                gl.dev_trim()
            except gl.GingerHipError as e:
                out["g2"][crv] = {"error": str(e)}

    # ---- NTT (single GPU per rank; reported from rank 0)
    if not args.no_ntt and rank == 0:
        N = 1 << args.ntt_log_n
        a = S.random_scalars_np(N, seed=7, below=pyref.P6.p)
        buf = gl.DeviceBuffer(N * 96).upload(a)
        dom = gl.EvaluationDomain("mnt4753_fr", N)
        for _ in range(max(1, args.warmup)):
            dom.fft_dev(buf, 0)
        ks, walls = [], []
        for i in range(max(3, args.steps)):
            t1 = time.perf_counter()
            dom.fft_dev(buf, i & 3)          # cycle fft / ifft / coset_fft / coset_ifft
            walls.append((time.perf_counter() - t1) * 1e3)
            ks.append(gl.fft_last_kernel_ms())
        buf.free()
        t1 = time.perf_counter()
        dom.fft(a)                               # gh_fft: host vector in, host vector out (pageable), PCIe both ways
        h2h_ms = (time.perf_counter() - t1) * 1e3
        ntt_ms = float(np.mean(ks))
        nb = 2.0 * N * 96
        out["ntt"] = {"field": "MNT4-753 Fr", "log_n": args.ntt_log_n, "ms": ntt_ms, "wall_ms": float(np.mean(walls)),
                      "transforms": "fft, ifft, coset_fft, coset_ifft cycled; device resident, in place",
                      "host_to_host_ms": h2h_ms, "host_to_host_note": "gh_fft from / to pageable host memory (PCIe both ways), one call; never the headline",
                      "roofline": {"kernel": "ntt_pass_kernel<P6> (all passes of one transform)", "bound": "hbm",
                                   "achieved": nb / (ntt_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": nb / (ntt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic_ntt,
                                   "algorithmic_bytes_per_transform": nb}}

    # ---- CPU baseline: the oracle (restated reference algorithm, C++) on the host cores, bounded sample
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        cores = os.cpu_count() or 1
        m = min(n, 1 << 16)        # BASELINE configs[0]: 2^16 pairs on the CPU path
        import math
        c_ref = 3 if m < 32 else math.ceil(2.0 / 3.0 * math.log2(m) + 2.0)      # variable_base.rs:14-18
        msm_threads = min(cores, -(-753 // c_ref))                               # one task per window (:30-31)
        bases = rb.download(0, m)                                                # the first m resident bases, Montgomery rows
        t1 = time.perf_counter()
        exp = S.oracle_msm(curve, bases[:m], None, scalars[:m], msm_threads)
        cpu_s = time.perf_counter() - t1
        got = rb.msm(scalars[:m])
        g_xy, g_inf = gl.proj_to_affine(curve, got)
        e_xy, e_inf = S.oracle_affine(curve, exp)
        parity = bool(g_inf == e_inf and (g_xy == e_xy).all())
        cpu_model = ""
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        m1 = min(m, 1 << 12)       # the same algorithm on ONE core, smaller sample (c = 10 at 2^12, 76 windows in sequence)
        t1 = time.perf_counter()
        S.oracle_msm(curve, bases[:m1], None, scalars[:m1], 1)
        cpu1_s = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": m / cpu_s, "unit": "scalar-muls/s", "cores": msm_threads, "host_cores": cores, "kind": "port",
                               "cpu_model": cpu_model,
                               "single_core": {"value": m1 / cpu1_s, "unit": "scalar-muls/s", "cores": 1,
                                               "sample": "first 2^%d pairs, %.2f s" % (int(np.log2(m1)), cpu1_s)},
                               "sample": "oracle (C++ restatement of variable_base.rs:10-83, window-parallel) on the first 2^%d pairs of the same inputs, %.2f s" % (int(np.log2(m)), cpu_s),
                               "gpu_matches_oracle_on_sample": parity}
        if not parity:
            out["error"] = "GPU result differs from the oracle on the CPU-baseline sample"
        if "ntt" in out:
            ln = min(args.ntt_log_n, 20)
            a = S.random_scalars_np(1 << ln, seed=8, below=pyref.P6.p)
            t1 = time.perf_counter()
            fft_threads = min(cores, 32)     # best_fft splits into 2^floor(log2 threads) sub-FFTs and its O(n*P) gather grows with P
            ref = S.oracle_fft("mnt4753_fr", a, ln, 0, fft_threads)
            cpu_ms = (time.perf_counter() - t1) * 1e3
            got = gl.EvaluationDomain("mnt4753_fr", 1 << ln).fft(a).reshape(-1, 12)
            out["ntt"]["cpu_baseline"] = {"log_n": ln, "ms": cpu_ms, "cores": fft_threads, "host_cores": cores, "kind": "port",
                                          "sample": "oracle best_fft (domain.rs:305-416) at 2^%d" % ln,
                                          "gpu_ms_same_size": gl.fft_last_kernel_ms(),
                                          "gpu_matches_oracle_on_sample": bool((got == ref).all())}
    rb.free()
    ds.free()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1:
        cdist.shutdown()
        dist.destroy_process_group()
    if out.get("error"):
        sys.exit(1)


if __name__ == "__main__":
    main()
