#!/usr/bin/env python3
"""Turn what tools/gpu_round.sh left under gpurun_out/ into the committed artifacts of a round:

    python tools/collect_round.py r01i

writes profiles/<tag>_bench_<workload>.json (the bench lines), profiles/<tag>_<workload>_rocprofv3.txt (kernel stats +
PMC passes, via tools/rocpd_summary.py) and profiles/hbm_traffic.json (HBM bytes per launch of each workload's
dominant kernel: FETCH_SIZE x 2 -- the gfx950 correction of MI355X_MICROARCH.md, HBM section -- + WRITE_SIZE, in KiB).
Removes the same files of older tags of the same round (r02a when r02b arrives; r01* stay).  The rocpd databases of a round exceed what gpurun copies back, so gpu_round.sh
runs this on the GPU box with --stage (artifacts are copied to gpurun_out/profiles_<tag>/) and deletes the databases;
locally: cp gpurun_out/profiles_<tag>/* profiles/ (and git rm the previous tag's files)."""
import io
import json
import sqlite3
import sys
from contextlib import redirect_stdout
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import rocpd_summary  # noqa: E402

WORKLOADS = ["aac", "mp3", "vorbis", "flac", "alac", "mp3q", "vorbisf", "aacjs", "aactns", "flacp", "alacp"]


def pmc_avg(db, kernel, counter):
    c = sqlite3.connect(db)
    row = c.execute("select avg(value) from counters_collection where kernel_name like ? and counter_name = ?",
                    ("%" + kernel + "%", counter)).fetchone()
    return None if row is None or row[0] is None else float(row[0])


KERNELS = {"aac": "aac_synth_quad_kernel", "mp3": "mp3_synth_kernel", "vorbis": "vorbis_synth_wave_kernel",
           "flac": "flac_restore_f64_kernel", "mp3q": "mp3_synth_kernel<4, true>", "vorbisf": "vorbis_synth_wave_kernel<2>", "aacjs": "aac_synth_quad_kernel<true>", "aactns": "aac_synth_quad_kernel<true>",
           # (four instantiations are launched per step and three return at once: count the one the workload's wavefronts run in)
           "alac": "alac_predict_kernel<false, true, true>", "flacp": "flac_restore_f64_kernel", "alacp": "alac_predict_kernel<false, true, true>"}


def bench_line(path):
    for line in reversed(path.read_text().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line), line
    raise SystemExit("no bench line in %s" % path)


def main(tag, traffic_only=False):
    out, prof = ROOT / "gpurun_out", ROOT / "profiles"
    traffic = {"_note": "HBM traffic per launch of each workload's dominant kernel from rocprofv3 PMC passes (separate --pmc "
                        "FETCH_SIZE / WRITE_SIZE runs of `bench.py --workload W --steps 3 --warmup 1 --no-cpu-baseline`, tools/gpu_round.sh). "
                        "FETCH_SIZE is doubled per the gfx950 correction in MI355X_MICROARCH.md (HBM section), calibrated in round 1 on "
                        "torch's exp2 kernel (32 MiB read reports 16 400 KB). bench.py copies bytes_per_launch into roofline.traffic."}
    if not traffic_only:
        for old in list(prof.glob("r0*_bench_*.json")) + list(prof.glob("r0*_rocprofv3.txt")):
            if old.name.startswith(tag[:3]) and not old.name.startswith(tag + "_"):  # older tags of the SAME round only
                old.unlink()
    for w in WORKLOADS:
        if traffic_only:  # on the GPU box, before the bench lines exist: the PMC run's own bench line has the byte count
            bench, _ = bench_line(out / ("pmc_%s_%s_FETCH_SIZE.log" % (tag, w)))
        else:
            bench, line = bench_line(out / ("bench_%s.json" % w))
            (prof / ("%s_bench_%s.json" % (tag, w))).write_text(line + "\n")
        dbs = [out / ("prof_%s_%s" % (tag, w)) / ("%s_results.db" % w)]
        pm = {}
        for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
            db = out / ("pmc_%s_%s_%s" % (tag, w, cnt)) / ("%s_results.db" % w)
            if db.exists():
                dbs.append(db)
                pm[cnt] = pmc_avg(str(db), KERNELS[w], cnt)
        if not traffic_only:
            buf = io.StringIO()
            with redirect_stdout(buf):
                rocpd_summary.main([str(d.relative_to(ROOT)) for d in dbs if d.exists()])
            (prof / ("%s_%s_rocprofv3.txt" % (tag, w))).write_text(buf.getvalue())
        if pm.get("FETCH_SIZE") is not None and pm.get("WRITE_SIZE") is not None:
            traffic[w] = {"kernel": KERNELS[w], "fetch_size_kb_raw": pm["FETCH_SIZE"], "write_size_kb": pm["WRITE_SIZE"],
                          "bytes_per_launch": int(round((2 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024)),
                          "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
                          # (the library the counters were taken on: bench.py flags a line whose library differs)
                          "source_sha256": ((bench.get("library") or {}).get("build") or {}).get("source_sha256"),
                          "source": "profiles/%s_%s_rocprofv3.txt" % (tag, w)}
            print(w, "traffic / algorithmic = %.4f" % (traffic[w]["bytes_per_launch"] / traffic[w]["algorithmic_bytes_per_launch"]))
    (prof / "hbm_traffic.json").write_text(json.dumps(traffic, indent=1) + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01", traffic_only="--traffic-only" in sys.argv)
    if "--stage" in sys.argv:  # on the GPU box: leave the artifacts under gpurun_out/ (the only directory copied back)
        import shutil
        stage = ROOT / "gpurun_out" / ("profiles_" + sys.argv[1])
        stage.mkdir(exist_ok=True)
        for f in list((ROOT / "profiles").glob(sys.argv[1] + "_*")) + [ROOT / "profiles" / "hbm_traffic.json"]:
            shutil.copy(f, stage / f.name)
