#!/usr/bin/env python3
"""HBM traffic per launch of the step kernels from rocprofv3 PMC passes -> profiles/step_traffic.json.

Run ON the GPU box (from anywhere):  python3 tools/pmc_traffic.py gpurun_out/<tag>_traffic
For every workload bench.py quotes a `traffic` for, two separate passes of tools/prof_step.py are taken, one with
`--pmc FETCH_SIZE` and one with `--pmc WRITE_SIZE` (they cannot share a pass on gfx950; no trace domain is combined
with them), and the per-dispatch averages of the workload's step kernel are corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters are in KiB; FETCH_SIZE is doubled (gfx950 tallies the
128-byte requests of wide coalesced reads at 64 bytes); WRITE_SIZE is exact for 16-byte-per-lane streaming stores.
The file carries `_lib.source_hash()` of the sources the measured library was built from: bench.py quotes a figure
only while that hash is the running library's.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORKLOADS = [  # (workload name as bench.py spells it, prof_step.py arguments)
    ("overcooked cramped_room 32768", ["--game", "overcooked", "--layout", "cramped_room", "--worlds", "32768"]),
    ("overcooked asymmetric_advantages 32768", ["--game", "overcooked", "--layout", "asymmetric_advantages", "--worlds", "32768"]),
    ("overcooked coordination_ring 32768", ["--game", "overcooked", "--layout", "coordination_ring", "--worlds", "32768"]),
    ("overcooked forced_coordination 32768", ["--game", "overcooked", "--layout", "forced_coordination", "--worlds", "32768"]),
    ("overcooked counter_circuit 32768", ["--game", "overcooked", "--layout", "counter_circuit", "--worlds", "32768"]),
    ("hanabi 65536", ["--game", "hanabi", "--worlds", "65536"]),
    ("cartpole 1048576", ["--game", "cartpole", "--worlds", "1048576"]),
    ("simplecooked simple 32768", ["--game", "simplecooked", "--worlds", "32768"]),
    ("balance_beam 1048576", ["--game", "balance", "--worlds", "1048576"]),
]


def one_pass(counter, prof_args, out_dir, log):
    cmd = ["timeout", "-k", "10", "150", "rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", out_dir, "-o", "p", "--",
           "python3", os.path.join(ROOT, "tools", "prof_step.py")] + prof_args + ["--steps", str(STEPS)]
    env = dict(os.environ, TMPDIR="/tmp")
    proc = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    open(log, "w").write(proc.stdout + proc.stderr)
    if proc.returncode != 0:
        raise SystemExit(f"rocprofv3 pass failed ({counter}, {prof_args}): see {log}")
    done = [ln for ln in proc.stdout.splitlines() if ln.startswith("done ")]
    # "done <game> <worlds> <steps> <kernel name ...> <bytes per world-step>"
    return " ".join(done[-1].split()[4:-1]) if done else None


STEPS = 30  # dispatches measured per pass: the last STEPS of the kernel (prof_step.py may run warm-up steps of the same kernel first)


def per_dispatch(out_dir, counter, kernel):
    acc = collections.defaultdict(list)
    for f in glob.glob(out_dir + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and kernel in row["Kernel_Name"]:
                acc[row["Kernel_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    if not acc:
        return None, 0
    name, rows = max(acc.items(), key=lambda kv: len(kv[1]))
    last = [v for _, v in sorted(rows)[-STEPS:]]
    return sum(last) / len(last), len(last)


def main():
    out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "traffic"))
    os.makedirs(out, exist_ok=True)
    from madrona_rl_envs_playground_amd import _lib
    rows = []
    for k, (workload, prof_args) in enumerate(WORKLOADS):
        kernel = None
        vals = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, f"w{k}_{counter}")
            kernel = one_pass(counter, prof_args, d, d + ".log") or kernel
            vals[counter] = per_dispatch(d, counter, kernel.split("<")[0] if kernel else "mrl_")
        (fetch_kib, nf), (write_kib, nw) = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
        if fetch_kib is None or write_kib is None:
            print(f"{workload}: no rows for {kernel}", flush=True)
            continue
        traffic = int(round((2.0 * fetch_kib + write_kib) * 1024))
        rows.append({"workload": workload, "kernel": kernel, "FETCH_SIZE_KiB_avg": fetch_kib, "WRITE_SIZE_KiB_avg": write_kib,
                     "dispatches": min(nf, nw), "traffic_bytes_per_launch": traffic})
        print(f"{workload}: {kernel}  fetch {fetch_kib:.1f} KiB x2 + write {write_kib:.1f} KiB = {traffic / 1e6:.2f} MB per launch", flush=True)
    doc = {"csrc_sha16": _lib.build_hash(),
           "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of tools/prof_step.py (30 dispatches each); both in "
                   "KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-byte read requests at 64 bytes), WRITE_SIZE exact "
                   "for 16-byte streaming stores.  Multi-launch steps (Cartpole / Hanabi two-launch path) list the step kernel only.",
           "launches": rows}
    json.dump(doc, open(os.path.join(out, "step_traffic.json"), "w"), indent=1)
    print("wrote", os.path.join(out, "step_traffic.json"))


if __name__ == "__main__":
    main()
