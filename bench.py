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
    for sub, ext in (("csrc", (".h", ".hip")), ("asmgen", (".py",))):       # hipcc sources and the generators of the assembly kernels
        d = os.path.join(ROOT, "ginger-lib_amd", sub)
        for f in sorted(os.listdir(d)):
            if f.endswith(ext):
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
    ap.add_argument("--no-prover", action="store_true", help="skip the `prover` object (BASELINE config 5: create_proof at 2^20 - 3 constraints)")
    ap.add_argument("--prover-log-n", type=int, default=20, help="domain size of the prover object's Benchmark circuit (constraints = 2^k - 3)")
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
            # Step 1, local and collective-free: can THIS rank load and bind librccl?  The ranks agree on the answer over gloo
            # BEFORE anyone enters ncclCommInitRank -- a rank that failed alone would otherwise leave the others blocked in it.
            local_ok = distmod.CDist.probe_rccl(gl)
            flag = torch.tensor([1 if local_ok else 0])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            err = "" if int(flag.item()) else ("librccl not loadable: %s" % gl.load_library().gh_last_error().decode() if not local_ok else "librccl not loadable on another rank")
            if not err:
                # Step 2, collective, under a watchdog: a communicator that does not come up within two minutes ends the job
                # with a non-zero status instead of hanging it.
                # (A timer THREAD: the main thread sits inside a ctypes call -- ncclCommInitRank or the probe all-gather -- where
                #  CPython runs no signal handler; ctypes releases the GIL, so the timer fires and ends the process.  ADVICE r3.)
                import threading

                def _watchdog():
                    sys.stderr.write("bench.py rank %d: RCCL communicator did not come up within 120 s\n" % rank)
                    sys.stderr.flush()
                    os._exit(3)
                wd = threading.Timer(120.0, _watchdog)
                wd.daemon = True
                wd.start()
                try:
                    cdist = distmod.CDist(gl, rank, world, transport="rccl", store=_get_default_store())
                    ident = np.zeros(36 * pyref.CURVES[args.curve].deg, dtype=np.uint64)
                    ident[12 * pyref.CURVES[args.curve].deg] = 1          # (0 : 1 : 0); any limbs do for the probe
                    cdist.allgather_fold(args.curve, ident)
                except Exception as e:      # noqa: reported below
                    err = "%s: %s" % (type(e).__name__, e)
                wd.cancel()
                flag = torch.tensor([0 if err else 1])
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0 and not err:
                    err = "RCCL failed on another rank"
            if err:
                sys.stderr.write("bench.py rank %d: RCCL exchange unavailable (%s); using the gloo callback transport\n" % (rank, err))
                if cdist is not None:
                    cdist.shutdown()
                cdist = _callback_transport()
                exchange_note = "gloo callback (RCCL unavailable: %s)" % err
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
        p0, step = Cc.mul(prng.next_u64() | 1, Cc.G), Cc.mul(prng.next_u64() | 1, Cc.G)
        xy, _ = S.bases_array(Cc, [p0, step])
        return gl.ResidentBases.chain(crv, xy[0], xy[1], count), p0, step

    def closed_form_ok(crv, p0, step, sc, xyz):
        """the MSM result against sum s_i (P_0 + i H) = (sum s_i) P_0 + (sum i s_i) H evaluated with Python integers
        (tests/support.py: no MSM code path involved) -- an independent answer at the full size"""
        Cc = pyref.CURVES[crv]
        e_xy, e_inf = S.affine_abi_of_point(Cc, S.chain_msm_closed_form(Cc, p0, step, sc))
        g_xy, g_inf = gl.proj_to_affine(crv, xyz)
        return bool(g_inf == e_inf and (np.asarray(g_xy).reshape(-1) == e_xy).all())

    t_key = time.perf_counter()
    rb, key_p0, key_step = chain_key(curve, n, 1 + rank)
    gl.load_library().gh_dev_sync()
    key_generate_s = time.perf_counter() - t_key
    scalars = S.random_scalars_np(n, seed=1000 + rank, below=C.order)
    ds = gl.DeviceBuffer(n * 96).upload(scalars)
    # The bases are a proving key: uploaded once, outside the timed region, and -- like the
    # layout conversion at upload -- expanded once into the shift table 2^(c w) P_i (gh_bases_precompute).
    # The per-window path (no table) is timed as well and reported under "per_window_path".
    plain = None
    table_info = None
    phases_alone = None
    if not args.no_precompute and not args.window:
        for _ in range(max(1, args.warmup)):
            rb.msm_dev(ds, n)
        gl.load_library().gh_dev_sync()
        PW_STEPS = 5
        t0 = time.perf_counter()
        for _ in range(PW_STEPS):
            plain_out = rb.msm_dev(ds, n)
        dt = (time.perf_counter() - t0) / PW_STEPS
        ptm = gl.msm_last_timing()
        t0 = time.perf_counter()
        gl.msm_batch_dev([(rb, ds, n)] * PW_STEPS)
        dtb = (time.perf_counter() - t0) / PW_STEPS
        plain = {"value": n / dt, "unit": "scalar-muls/s per GPU", "ms_per_step": dt * 1e3, "window_bits": ptm["window_bits"],
                 "num_windows": ptm["num_windows"], "accumulate_ms": ptm["accumulate_ms"], "steps": PW_STEPS,
                 "pipelined_batch": {"value": n / dtb, "ms_per_step": dtb * 1e3, "steps": PW_STEPS},
                 "closed_form_ok": closed_form_ok(curve, key_p0, key_step, scalars, plain_out)}
        t0 = time.perf_counter()
        c_tab = rb.precompute(0)
        rows = 752 // c_tab + 1
        table_info = {"window_bits": c_tab, "rows": rows, "bytes": rows * n * 208 * C.deg, "build_s": time.perf_counter() - t0,
                      "key_generate_s": key_generate_s,
                      "note": "one-time per resident key, outside the timed region (like the base upload); key_load_s = key_generate_s "
                              "(synthetic chain on the device; a real key pays its upload instead) + build_s"}
        table_info["key_load_s"] = key_generate_s + table_info["build_s"]
        a1 = gl.proj_to_affine(curve, rb.msm_dev(ds, n))
        t0 = time.perf_counter()
        solo_tm = []
        for _ in range(5):
            rb.msm_dev(ds, n)
            solo_tm.append(gl.msm_last_timing())
        table_info["single_msm_ms"] = (time.perf_counter() - t0) / 5 * 1e3    # latency of one MSM, nothing to overlap with
        table_info["single_msm_steps"] = 5
        phases_alone = {k: float(np.mean([t[k] for t in solo_tm])) for k in ("sort_ms", "accumulate_ms", "heavy_ms", "reduce_ms", "fold_ms")}
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
        # the partial sums of the whole batch travel in ONE all-gather (gh_partials_allgather_fold_batch), then k folds
        totals = cdist.allgather_fold_batch(curve, partials) if world > 1 else partials
        if world > 1:
            exch_us.append(cdist.last_exchange_us)
        own_partial[:] = [partials[-1]]
        return totals, tms

    exch_us = []
    own_partial = []

    def sync():
        gl.load_library().gh_dev_sync()      # the library's streams (this process holds no other GPU work)
        if world > 1:
            dist.barrier()

    # which accumulation the library's automatic policy picked for this curve (include/ginger_hip.h: gh_msm_set_affine)
    xyzz = C.deg == 1 and os.environ.get("GH_ACC_XYZZ", "1") != "0"
    AFF_MODE = "affine rounds (asmgen/g2_rounds.py) + projective finish" if os.environ.get("GH_AFF_ASM", "1") != "0" else "affine rounds (aff_kernels.h) + projective finish"
    bucket_mode = AFF_MODE if (C.deg > 1 and os.environ.get("GH_AFFINE", "2") != "0") or os.environ.get("GH_AFFINE") == "1" \
        else ("XYZZ mixed additions (madd-2008-s, 8 M + 2 S, Y3 as one dual product)" if xyzz else "projective mixed additions")
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
    per_add = (6 if bucket_mode.startswith("affine") else (10 if xyzz else 11)) * {1: 1, 2: 4, 3: 9}[C.deg]
    # every rank checks ITS partial sum (own key, own scalars) against the closed form; the line reports the AND over ranks
    cf_ok = closed_form_ok(curve, key_p0, key_step, scalars, own_partial[0])
    if world > 1:
        import torch
        f = torch.tensor([1 if cf_ok else 0])
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        cf_ok = bool(int(f.item()))
    fpmul_rate = madds * per_add / (acc_avg_ms * 1e-3)
    # measured on this box, in this run: the product peak and what the code object says about the three hot kernels
    peak_now, hot_kernels = None, None
    if rank == 0:
        try:
            peak_now = gl.measure_fpmul_peak()
            hot_kernels = {k: gl.kernel_resources(k) for k in ("g1_acc_p4", "g2_f2_bwd_r0", "g2_f3_bwd_r0", "ntt_p6_k8")}
            hot_kernels["note"] = ("generated gfx950 assembly (ginger-lib_amd/asmgen): G1 XYZZ bucket update, backward kernels of the Fq2 / Fq3 affine "
                                   "rounds, the 8-stage NTT pass; scratch_bytes_per_lane / registers / lds_bytes as hipFuncGetAttribute reports them for the loaded code object")
        except gl.GingerHipError as e:
            hot_kernels = {"error": str(e)}

    # HBM traffic per launch of the dominant kernels: PMC counters collected with rocprofv3 in separate
    # passes on this same command (profiles/r01_pmc_traffic.json); null if that file is absent or the
    # workload differs from the profiled one (2^20 pairs / 2^24 points).
    traffic_acc = traffic_ntt = None
    traffic_g2 = {}
    traffic_src = "profiles/r04_pmc_traffic.json"
    traffic_when = None
    try:
        tr = json.load(open(os.path.join(ROOT, traffic_src)))
        traffic_when = tr.get("measured")
        if tr.get("kernels_sha256") == kernels_sha():          # counters go stale when the kernels change: then null
            ent = tr.get("%s_2p%d" % (curve, n.bit_length() - 1))
            if ent and ent["window_bits"] == tm_last["window_bits"] and ent["bucket_sums"] == bucket_mode:
                traffic_acc = ent["fetch_bytes_per_msm"] + ent["write_bytes_per_msm"]
            for crv_g, lg_g in (("mnt4753_g2", 20), ("mnt6753_g2", 19)):
                eg = tr.get("%s_2p%d" % (crv_g, lg_g))
                if eg:
                    traffic_g2[crv_g] = eg["fetch_bytes_per_msm"] + eg["write_bytes_per_msm"]
            ent = tr.get("ntt_2p%d" % args.ntt_log_n)
            if ent:
                traffic_ntt = ent["fetch_bytes_per_transform"] + ent["write_bytes_per_transform"]
    except Exception:
        pass

    out = {
        "metric": "MNT4-753 G1 MSM scalar-muls/sec + 2^n NTT ms at 1/2/4/8 MI355X",
        "value": value,
        "unit": "scalar-muls/s",
        "n_gpus": world,
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
                   "resident_key_shift_table": table_info, "distinct_bases": n, "bucket_sums": bucket_mode,
                   "bucket_reduction": ("segment form (GH_REDUCE_LEAN=0)" if os.environ.get("GH_REDUCE_LEAN") == "0" else
                                        "lean (lane-level) form for every MSM of the batch but the last, from 2^25 list entries; segment form otherwise")
                                       if (C.deg == 1 and not args.no_pipeline) else "segment form",
                   "parallelism": "pairs sharded by rank, 1 all-gather of partial sums" if world > 1 else "single GPU"},
        "closed_form_ok": cf_ok,
        "closed_form_note": "each rank's partial sum of the last timed step == (sum s_i) P_0 + (sum i s_i) H on its chain key, evaluated with "
                            "Python integers (tests/support.py chain_msm_closed_form): an answer no MSM code path produced",
        "roofline": {"kernel": "%s (bucket accumulation of the %s MSM)" % (("gh_asm_acc_g1 (generated assembly of msm_accumulate_xyzz_kernel)" if os.environ.get("GH_ACC_ASM", "1") != "0" else "msm_accumulate_xyzz_kernel") if xyzz else "msm_accumulate_kernel / aff_round_kernel", curve), "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_acc,
                     "traffic_note": "FETCH_SIZE + WRITE_SIZE bytes of the accumulation launches of one MSM from %s (separate rocprofv3 --pmc passes), null when that file was measured on other kernel sources (sha256 of ginger-lib_amd/csrc and asmgen) or another workload; measured by the builder: %s" % (traffic_src, traffic_when or "box and date unknown"),
                     "avg_launch_ms": acc_avg_ms, "algorithmic_bytes_per_launch": alg_bytes,
                     "note": "integer-VALU bound by construction (SURVEY 8d): see valu"},
        "valu": {"achieved_fpmul_per_s": fpmul_rate, "peak_fpmul_per_s": FPMUL_PEAK_PER_S, "frac": fpmul_rate / FPMUL_PEAK_PER_S,
                 "peak_fpmul_per_s_measured_now": peak_now, "frac_of_measured_peak": (fpmul_rate / peak_now) if peak_now else None,
                 "fp_products_per_addition": per_add,
                 "note": "peak = the round-1 rr29 Montgomery-product microbenchmark (profiles/r01_microbench_valu_rates.txt); "
                         "peak_fpmul_per_s_measured_now = the same kind of loop (two interleaved products per lane, asmgen/microbench.py) "
                         "run by the library on THIS card in this process (gh_measure_fpmul_peak)"},
        "hot_kernels": hot_kernels,
        "phases_ms": phases,
        "phases_alone_ms": phases_alone,
        "phases_note": "per-MSM device phases by HIP events on their streams; with the pipelined batch the phases of neighbouring "
                       "steps overlap (sort and reduce run beside the previous / next accumulation) and are stretched by it, so they do not add "
                       "up to ms_per_step; phases_alone_ms = the same phases of ONE MSM with nothing beside it (mean of 5)",
        "pipelined": not args.no_pipeline,
    }
    if world > 1:
        out["exchange"] = {"transport": exchange_note or ("rccl all-gather behind gh_partials_allgather_fold" if backend == "rccl" else "gloo (rehearsal)"),
                           "ranks_seen_by_transport": cdist.world_seen, "rccl_ranks": cdist.rccl_ranks,
                           "rccl_ranks_note": "ncclCommCount of the live communicator; 0 = the exchange did NOT go over RCCL (callback transport)",
                           "rccl_version": cdist.rccl_version, "rccl_path": cdist.rccl_path,
                           "bytes_per_rank_and_msm": 288 * C.deg, "exchanges_per_timed_region": 1,
                           "avg_us": float(np.mean(exch_us[-1:])) if exch_us else None,
                           "note": "the partial sums of all timed steps travel in ONE all-gather after the batch; avg_us = that exchange"}
    if not cf_ok:
        out["error"] = "MSM result differs from the closed form on the chain key"
    if plain is not None:
        out["per_window_path"] = plain
        if not plain["same_affine_result_as_table_path"]:
            out["error"] = "table path and per-window path disagree"
        if not plain["closed_form_ok"]:
            out["error"] = "per-window path differs from the closed form"
    # ---- what the UNCHANGED prover would see through the Rust shim (rust/algebra-hip-sys + rust/patches): prover.rs:273-325
    #      issues its G1 MSMs one call after the other, each with (bases, scalars) host slices -> gh_msm_cached: the bases are
    #      hashed on every call (identity = content), found resident, and only the scalars cross PCIe; nothing is pipelined.
    #      Pageable host arrays, as a Vec is.  `handle_path` = the same four calls for a caller that holds the handle
    #      (gh_msm_resident: no hash).
    if rank == 0 and world == 1 and not args.window and not args.no_precompute:
        host_s = [scalars, scalars[::-1].copy(), scalars.copy(), scalars[::-1].copy()]
        host_bases = rb.download(0, n)
        gl.key_cache_clear()
        t0 = time.perf_counter()
        gl.msm_cached(curve, host_bases, host_s[0])              # first sighting: upload, per-window path (what gh_msm costs)
        first_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        gl.msm_cached(curve, host_bases, host_s[0])              # second sighting: the key gets its shift table
        second_ms = (time.perf_counter() - t0) * 1e3
        gl.msm_cached(curve, host_bases, host_s[0])
        t0 = time.perf_counter()
        outs4 = [gl.msm_cached(curve, host_bases, h) for h in host_s]
        dt4 = time.perf_counter() - t0
        kst = gl.key_cache_stats()
        gl.key_cache_clear()
        rb.msm(host_s[0])
        t0 = time.perf_counter()
        outs4h = [rb.msm(h) for h in host_s]
        dt4h = time.perf_counter() - t0
        out["drop_in_path"] = {"workload": "4 sequential gh_msm_cached calls with host (bases, scalars) slices: the G1 MSM sequence of one create_proof "
                                           "as the unchanged prover.rs:273-325 issues it through rust/algebra-hip-sys (bases hashed per call, "
                                           "resident after the first proof, shift table from the second sighting)",
                               "value": 4 * n / dt4, "unit": "scalar-muls/s", "ms_per_msm": dt4 / 4 * 1e3, "calls": 4, "pcie_inclusive": True,
                               "first_sighting_ms": first_ms, "second_sighting_ms_incl_table_build": second_ms,
                               "key_cache": kst,
                               "handle_path": {"value": 4 * n / dt4h, "ms_per_msm": dt4h / 4 * 1e3,
                                               "note": "gh_msm_resident on a caller-held handle: no packing, no hash"},
                               "closed_form_ok": closed_form_ok(curve, key_p0, key_step, host_s[0], outs4[0])
                                                 and closed_form_ok(curve, key_p0, key_step, host_s[1], outs4h[1])}
        if not out["drop_in_path"]["closed_form_ok"]:
            out["error"] = "drop-in path differs from the closed form"
        del host_bases

    # ---- BASELINE config 3's second size: 2^24 pairs (per-window path and shift table c = 23, 115 GB), same line
    if not args.no_2p24 and rank == 0 and world == 1 and curve == "mnt4753_g1" and not args.window and args.log_n < 24:
        try:
            n24 = 1 << 24
            rb24, p0_24, st_24 = chain_key(curve, n24, 77)
            s24 = S.random_scalars_np(n24, seed=2024, below=C.order)
            d24 = gl.DeviceBuffer(n24 * 96).upload(s24)
            K24 = 5
            rb24.msm_dev(d24, n24)
            t1 = time.perf_counter()
            for _ in range(K24):
                p_out = rb24.msm_dev(d24, n24)
            pw_ms = (time.perf_counter() - t1) * 1e3 / K24
            ptm = gl.msm_last_timing()
            t1 = time.perf_counter()
            c24 = rb24.precompute(0)
            build_s = time.perf_counter() - t1
            rb24.msm_dev(d24, n24)
            t1 = time.perf_counter()
            for _ in range(K24):
                t_out = rb24.msm_dev(d24, n24)
            tb_ms = (time.perf_counter() - t1) * 1e3 / K24
            ttm = gl.msm_last_timing()
            gl.msm_batch_dev([(rb24, d24, n24)] * 2)        # untimed: the second pipeline slot's buffers are allocated on first use
            t1 = time.perf_counter()
            b_out = gl.msm_batch_dev([(rb24, d24, n24)] * K24)
            bt_ms = (time.perf_counter() - t1) * 1e3 / K24
            # partial table: 8 rows at c = 21 (28 GB where the full table above is 115 GB) -- what a prover with several 2^24-base
            # queries can keep resident for all of them (gh_bases_precompute_rows)
            t1 = time.perf_counter()
            c24p = rb24.precompute(0, 8)
            pbuild_s = time.perf_counter() - t1
            rows24p = rb24.table_rows()
            rb24.msm_dev(d24, n24)
            t1 = time.perf_counter()
            for _ in range(K24):
                pt_out = rb24.msm_dev(d24, n24)
            ptb_ms = (time.perf_counter() - t1) * 1e3 / K24
            gl.msm_batch_dev([(rb24, d24, n24)] * 2)
            t1 = time.perf_counter()
            pb_out = gl.msm_batch_dev([(rb24, d24, n24)] * K24)
            pbt_ms = (time.perf_counter() - t1) * 1e3 / K24
            partial24 = {"value": n24 / ptb_ms * 1e3, "ms": ptb_ms, "window_bits": c24p, "rows": rows24p, "bytes": rows24p * n24 * 208,
                         "build_s": pbuild_s, "pipelined_batch": {"value": n24 / pbt_ms * 1e3, "ms_per_msm": pbt_ms, "steps": K24},
                         "closed_form_ok": closed_form_ok(curve, p0_24, st_24, s24, pt_out) and closed_form_ok(curve, p0_24, st_24, s24, pb_out[-1])}
            a0, a1 = gl.proj_to_affine(curve, p_out), gl.proj_to_affine(curve, t_out)
            cf24 = closed_form_ok(curve, p0_24, st_24, s24, b_out[-1]) and closed_form_ok(curve, p0_24, st_24, s24, p_out)
            del s24
            out["msm_2p24"] = {
                "workload": "MNT4-753 G1 VariableBaseMSM, 2^24 distinct pairs, resident",
                "per_window_path": {"value": n24 / pw_ms * 1e3, "ms": pw_ms, "window_bits": ptm["window_bits"], "num_windows": ptm["num_windows"],
                                    "accumulate_ms": ptm["accumulate_ms"]},
                "shift_table": {"value": n24 / tb_ms * 1e3, "ms": tb_ms, "window_bits": c24, "rows": 752 // c24 + 1,
                                "bytes": (752 // c24 + 1) * n24 * 208, "build_s": build_s, "accumulate_ms": ttm["accumulate_ms"],
                                "pipelined_batch": {"value": n24 / bt_ms * 1e3, "ms_per_msm": bt_ms, "steps": K24}},
                "partial_table": partial24,
                "steps_per_figure": K24, "closed_form_ok": cf24 and partial24["closed_form_ok"],
                "roofline": {"bound": "hbm", "achieved": 288.0 * n24 / (ttm["accumulate_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": 288.0 * n24 / (ttm["accumulate_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                             "avg_launch_ms": ttm["accumulate_ms"], "algorithmic_bytes_per_launch": 288.0 * n24},
                "valu": {"achieved_fpmul_per_s": ttm["accumulate_madds"] * per_add / (ttm["accumulate_ms"] * 1e-3), "peak_fpmul_per_s": FPMUL_PEAK_PER_S,
                         "frac": ttm["accumulate_madds"] * per_add / (ttm["accumulate_ms"] * 1e-3) / FPMUL_PEAK_PER_S},
                "unit": "scalar-muls/s", "same_affine_result": bool(a0[1] == a1[1] and (a0[0] == a1[0]).all())}
            if not out["msm_2p24"]["same_affine_result"]:
                out["error"] = "2^24: table path and per-window path disagree"
            if not cf24:
                out["error"] = "2^24: result differs from the closed form"
            d24.free()
            rb24.free()
            gl.dev_trim()
        except gl.GingerHipError as e:          # e.g. not enough HBM next to another tenant: the object is absent, the line stands
            out["msm_2p24"] = {"error": str(e)}

    # ---- G2 (the prover's b_g2 MSM: BASELINE configs 4 / 5 shapes), one MSM at a time and as a pipelined batch of ten
    if not args.no_g2 and rank == 0 and world == 1 and curve == "mnt4753_g1" and not args.window:
        out["g2"] = {}
        for crv, lg in (("mnt4753_g2", 20), ("mnt6753_g2", 19)):
            try:
                ng = 1 << lg
                Cg = pyref.CURVES[crv]
                rbg, p0g, stg = chain_key(crv, ng, 31)
                sg = S.random_scalars_np(ng, seed=4242, below=Cg.order)
                dg = gl.DeviceBuffer(ng * 96).upload(sg)
                t1 = time.perf_counter()
                cg = rbg.precompute(0)
                build_s = time.perf_counter() - t1
                rbg.msm_dev(dg, ng)
                KG = 5
                t1 = time.perf_counter()
                for _ in range(KG):
                    g_out = rbg.msm_dev(dg, ng)
                one_ms = (time.perf_counter() - t1) / KG * 1e3
                gtm = gl.msm_last_timing()
                gl.msm_batch_dev([(rbg, dg, ng)] * 2)            # untimed: the second pipeline slot's buffers are allocated on first use
                KB = 10                                          # MSMs in the timed batch (its first sort and last reduction are not hidden)
                t1 = time.perf_counter()
                gb_out = gl.msm_batch_dev([(rbg, dg, ng)] * KB)
                bt_ms = (time.perf_counter() - t1) * 1e3 / KB
                g_bytes = (192.0 * Cg.deg + 96.0) * ng
                g_affine = os.environ.get("GH_AFFINE", "2") != "0"
                g_per_add = (6 if g_affine else 11) * {2: 4, 3: 9}[Cg.deg]
                out["g2"][crv] = {"workload": "%s VariableBaseMSM, 2^%d pairs, resident key with shift table" % (crv, lg),
                                  "steps_per_figure": KG, "steps_in_pipelined_batch": KB,
                                  "closed_form_ok": closed_form_ok(crv, p0g, stg, sg, g_out) and closed_form_ok(crv, p0g, stg, sg, gb_out[-1]),
                                  "roofline": {"kernel": "gh_asm_aff_* round kernels + inversion + projective finish (all accumulation launches of one MSM)", "bound": "hbm",
                                               "achieved": g_bytes / (gtm["accumulate_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                               "frac": g_bytes / (gtm["accumulate_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic_g2.get(crv),
                                               "avg_launch_ms": gtm["accumulate_ms"], "algorithmic_bytes_per_launch": g_bytes},
                                  "valu": {"achieved_fpmul_per_s": gtm["accumulate_madds"] * g_per_add / (gtm["accumulate_ms"] * 1e-3),
                                           "peak_fpmul_per_s": FPMUL_PEAK_PER_S, "fp_products_per_addition": g_per_add,
                                           "frac": gtm["accumulate_madds"] * g_per_add / (gtm["accumulate_ms"] * 1e-3) / FPMUL_PEAK_PER_S},
                                  "value": ng / bt_ms * 1e3, "unit": "scalar-muls/s", "ms_per_msm_pipelined": bt_ms, "single_msm_ms": one_ms,
                                  "window_bits": cg, "table_build_s": build_s, "bucket_sums": AFF_MODE if os.environ.get("GH_AFFINE", "2") != "0" else "projective mixed additions",
                                  "phases_ms": {k: gtm[k] for k in ("sort_ms", "accumulate_ms", "reduce_ms", "fold_ms") if k in gtm}}
                if not out["g2"][crv]["closed_form_ok"]:
                    out["error"] = "%s: result differs from the closed form" % crv
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
                      # the passes are bound by the issue rate of the 753-bit products, not by HBM: log_n / 2 products per element on average
                      # over the four kinds (a butterfly = one product, the inter-pass twiddle folded in: asmgen/ntt_pass.py)
                      "valu": {"fp_products_per_element": args.ntt_log_n / 2.0,
                               "achieved_fpmul_per_s": N * (args.ntt_log_n / 2.0) / (ntt_ms * 1e-3),
                               "peak_fpmul_per_s_measured_now": peak_now,
                               "frac_of_measured_peak": (N * (args.ntt_log_n / 2.0) / (ntt_ms * 1e-3) / peak_now) if peak_now else None},
                      "roofline": {"kernel": "gh_asm_ntt_p6_k* (generated assembly; all passes of one transform)", "bound": "hbm",
                                   "achieved": nb / (ntt_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": nb / (ntt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic_ntt,
                                   "traffic_note": "raw FETCH_SIZE + WRITE_SIZE; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-reports wide coalesced "
                                                   "streaming reads by 2x, so the true read traffic of these passes is up to twice the fetch part "
                                                   "(profiles/r03_pmc_traffic.json holds fetch and write separately)",
                                   "algorithmic_bytes_per_transform": nb}}

    # ---- CPU baseline: the oracle (restated reference algorithm, C++) on the host cores, bounded sample
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        cores = os.cpu_count() or 1
        m = n                      # the headline's own size (2^20 pairs: ~11 s on the box's host cores)
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
        m1 = min(m, 1 << 13)       # the same algorithm on ONE core, smaller sample (c = 11 at 2^13, 69 windows in sequence)
        t1 = time.perf_counter()
        S.oracle_msm(curve, bases[:m1], None, scalars[:m1], 1)
        cpu1_s = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": m / cpu_s, "unit": "scalar-muls/s", "cores": msm_threads, "host_cores": cores, "kind": "port",
                               "cpu_model": cpu_model,
                               "single_core": {"value": m1 / cpu1_s, "unit": "scalar-muls/s", "cores": 1,
                                               "sample": "first 2^%d pairs, %.2f s" % (int(np.log2(m1)), cpu1_s)},
                               "sample": "oracle (C++ restatement of variable_base.rs:10-83, one task per window as the reference: %d windows at c = %d) on %s of the same inputs, %.2f s" % (
                                   -(-753 // c_ref), c_ref, "ALL 2^%d pairs" % int(np.log2(m)) if m == n else "the first 2^%d pairs" % int(np.log2(m)), cpu_s),
                               "gpu_matches_oracle_on_sample": parity}
        if not parity:
            out["error"] = "GPU result differs from the oracle on the CPU-baseline sample"
        if "ntt" in out:
            ln = min(args.ntt_log_n, 24)       # the GPU figure's own size (2^24: ~20-30 s of best_fft on 32 threads)
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
    # ---- BASELINE config 5: create_proof of the `Benchmark` circuit (examples/snark-scalability/constraints.rs:20-92) with
    #      2^20 - 3 constraints over MNT4-753, from a key GENERATED on the device (groth16.generate_parameters mirrors
    #      generator.rs:146-335: gh_lagrange_coefficients + gh_fixed_base_msm_affine), serialised as Parameters::write does, parsed
    #      and made resident (gh_bases_upload_wire + shift tables); the timed part is prover.rs:201-346 from the host's limb arrays
    #      on: rows over PCIe, witness map, into_repr, four G1 MSMs as one batch and one G2 MSM, host fold.  A, B, C are checked
    #      against the Groth16 equations evaluated in the exponent from the toxic waste (no MSM / FFT code path involved).
    if not args.no_prover and rank == 0 and world == 1 and curve == "mnt4753_g1" and not args.window:
        try:
            gl.dev_trim()
            import importlib
            groth16 = importlib.import_module("ginger_lib_amd.groth16")
            def wire(Cc, P):          # GroupAffine::write (short_weierstrass_projective.rs:185-192) of a Python point
                xs, ys = (tuple([0] * Cc.deg), tuple([1] + [0] * (Cc.deg - 1))) if P is None else P
                return b"".join(int(v).to_bytes(96, "little") for v in tuple(xs) + tuple(ys)) + (b"\x01" if P is None else b"\x00")
            pairing = "mnt4753"
            C1, C2 = pyref.CURVES[pairing + "_g1"], pyref.CURVES[pairing + "_g2"]
            rr = C1.order
            n_con = (1 << args.prover_log_n) - 3
            prng = pyref.Rng(2026)
            alpha, beta, gamma, delta, tau, r_, s_ = (prng.field_elem(rr) for _ in range(7))
            g1, g2 = C1.mul(prng.next_u64() | 1, C1.G), C2.mul(prng.next_u64() | 1, C2.G)
            t1 = time.perf_counter()
            lcs = groth16.benchmark_circuit_lcs(n_con)
            blob, info = groth16.generate_parameters(gl, pairing, lcs, alpha, beta, gamma, delta, tau, S.proj_array(C1, g1), S.proj_array(C2, g2))
            gen_s = time.perf_counter() - t1
            del lcs
            t1 = time.perf_counter()
            key = groth16.ResidentProvingKey.from_parameters(gl, pairing, blob, 3)
            load_s = time.perf_counter() - t1
            blob_mb = len(blob) / 1048576.0
            del blob
            try:
                t1 = time.perf_counter()
                rows = groth16.benchmark_circuit_rows(pairing, n_con)
                prep = key.prepare_rows(rows, 0, 0, 0)
                host_rows_s = time.perf_counter() - t1
                key.prove_prepared(prep, r_, s_)                       # untimed: pipeline slots and pools are allocated on first use
                KP, tms, proof = 3, [], None
                for _ in range(KP):
                    tm = {}
                    t1 = time.perf_counter()
                    proof = key.prove_prepared(prep, r_, s_, timing=tm)
                    tm["proof_ms"] = (time.perf_counter() - t1) * 1e3
                    tms.append(tm)
            finally:
                key.free()
                gl.dev_trim()
            a_, b_, c_, l_, zt_ = info["qap"]
            asg = rows[1]
            sa = sum(x * y for x, y in zip(asg, a_)) % rr
            sb = sum(x * y for x, y in zip(asg, b_)) % rr
            sc = sum(x * y for x, y in zip(asg, c_)) % rr
            di = pow(delta, -1, rr)
            A_s = (alpha + sa + r_ * delta) % rr
            B_s = (beta + sb + s_ * delta) % rr
            H_s = (sa * sb - sc) * di % rr                           # h(t) Z(t) / delta with d1 = d2 = d3 = 0 (QAP divisibility)
            C_s = (sum(asg[i] * l_[i] for i in range(3, len(asg))) + H_s + s_ * A_s + r_ * B_s - r_ * s_ * delta) % rr
            exp = wire(C1, C1.mul(A_s, g1)) + wire(C2, C2.mul(B_s, g2)) + wire(C1, C1.mul(C_s, g1))
            mean = lambda k: float(np.mean([t[k] for t in tms]))
            out["prover"] = {"workload": "Groth16 create_proof, MNT4-753, `Benchmark` circuit with 2^%d - 3 constraints (BASELINE config 5), key generated on the "
                                         "device, resident with shift tables" % args.prover_log_n,
                             "proofs_timed": KP, "proof_ms": mean("proof_ms"), "rows_upload_ms": mean("rows_upload_ms"),
                             "witness_map_ms": mean("witness_map_ms"), "msm_stage_ms": mean("msm_stage_ms"),
                             "device_ms": mean("witness_map_ms") + mean("msm_stage_ms"),
                             "stages": "prover.rs:201-346 from the host's limb arrays: rows over PCIe (3 x 2^%d x 96 B), gh_witness_map_dev + into_repr, "
                                       "4 G1 MSMs as one pipelined batch + 1 G2 MSM + host fold; the Python-integer row evaluation (host_rows_s) is "
                                       "outside, as the reference's constraint synthesis is" % args.prover_log_n,
                             "key_generate_s": gen_s, "key_generate_fixed_base_s": info["fixed_base"]["fixed_base_s"],
                             "key_bytes_mb": blob_mb, "key_load_s": load_s, "host_rows_s": host_rows_s,
                             "closed_form_ok": bool(proof == exp),
                             "closed_form_note": "Proof::write bytes equal (alpha + sum a_i(t) x_i + r delta) g1, (beta + sum b_i(t) x_i + s delta) g2, "
                                                 "(sum_aux x_i l_i + h(t) Z(t) / delta + s A + r B - r s delta) g1 computed from the toxic waste with Python integers"}
            if not out["prover"]["closed_form_ok"]:
                out["error"] = "prover: proof differs from the Groth16 equations in the exponent"
        except gl.GingerHipError as e:
            out["prover"] = {"error": str(e)}
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1:
        cdist.shutdown()
        dist.destroy_process_group()
    if out.get("error"):
        sys.exit(1)


if __name__ == "__main__":
    main()
