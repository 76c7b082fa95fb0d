#!/usr/bin/env python3
"""Condense rocprofv3 output into the small files committed under profiles/.

usage: summarize_profile.py TAG KERNEL_STATS_CSV [FETCH_COUNTER_CSV WRITE_COUNTER_CSV [WORKLOAD]]

WORKLOAD names what was profiled in bench.py's notation (default nested-c3-512x512-b16-exact); together with the
source hash of the library that ran (unet_amd._lib.source_hash(), the same tree this script is run from on the GPU
box) it lets bench.py refuse HBM-traffic numbers that were recorded for another build or workload.

Writes profiles/TAG_kernel_stats.csv (copy of --kernel-trace --stats) and, when the two
`--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes are given, profiles/TAG_pmc_traffic.json with per-kernel
HBM traffic per launch.  Units/corrections follow MI355X_MICROARCH.md §HBM: both counters are in KiB;
on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so reads = 2 x FETCH_SIZE
(checked here on convert_input_kernel: 2 x 24.0 MiB vs 48.0 MiB of fp32 input actually read)."""
import collections, csv, json, os, re, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(name):
    """'void unetpp::conv3x3_bias_relu_kernel<2, 16, ...>(unetpp::ConvArgs)' or a mangled name -> 'conv3x3_bias_relu_kernel<2, 16, ...>'"""
    if name.startswith("_Z"):
        for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
            try:
                d = subprocess.run([tool, name], capture_output=True, text=True).stdout.strip()
            except OSError:
                continue
            if d and not d.startswith("_Z"):
                name = d
                break
    m = re.match(r"_ZN6unetpp\d+([A-Za-z0-9_]+?)ILi(\d+)EEE", name)      # llvm-cxxfilt does not know _Float16 (DF16_)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name.replace("unetpp::", "")


def agg(path, counter):
    d = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = short(r["Kernel_Name"]); d[k][0] += float(r["Counter_Value"]); d[k][1] += 1
    return d


def main():
    tag, stats = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, "profiles")
    shutil.copy(stats, os.path.join(out, f"{tag}_kernel_stats.csv"))
    if len(sys.argv) >= 5:
        f, w = agg(sys.argv[3], "FETCH_SIZE"), agg(sys.argv[4], "WRITE_SIZE")
        res = {}
        for k in f:
            fk, n = f[k]; wk, n2 = w.get(k, [0.0, 1])
            res[k] = {"launches_sampled": n, "fetch_size_kib_per_launch_raw": fk / n,
                      "read_bytes_per_launch": 2.0 * fk / n * 1024, "write_bytes_per_launch": wk / max(n2, 1) * 1024,
                      "hbm_bytes_per_launch": (2.0 * fk / n + wk / max(n2, 1)) * 1024}
        from unet_amd import _lib
        workload = sys.argv[5] if len(sys.argv) >= 6 else "nested-c3-512x512-b16-exact"
        json.dump({"src_hash": _lib.source_hash(), "workload": workload, "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KiB units; reads = 2 x FETCH_SIZE "
                           "(gfx950 wide-read correction, MI355X_MICROARCH.md §HBM); averages over all launches of a kernel "
                           "(layers of different shapes share a kernel instantiation)", "kernels": res},
                  open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1)
        for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:8]:
            print(f"{k:70s} {v['hbm_bytes_per_launch']/1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main()
