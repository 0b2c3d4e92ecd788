"""Refresh profiles/roofline_traffic.json from profiles/<tag>_hbm_counters.json (tools/summarize_prof.py) for the kernel the
bench quotes its roofline on, and record what the counters were taken ON: the commit and the SHA-256 of the kernel's source
file -- bench.py quotes the figure only while that file is unchanged (the GPU box has no .git to ask).
usage: python tools/update_roofline_traffic.py <tag> [n p lanes]"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
tag = sys.argv[1]
n, p, lanes = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (100_000, 5_000, 18)
src = "sparse-lm_amd/csrc/split_kernels.hpp"
# the kernel of the headline's passes: sixteen lanes on the matrix cores, up to four more on the vector units beside them
kernel = ("slm::xtr_mfma_kernel(slm::SplitArgs)" if lanes <= 16 else "slm::xtr18_mfma_kernel(slm::SplitArgs)" if lanes <= 18 else
          "slm::xtr20_mfma_kernel(slm::SplitArgs)" if lanes <= 20 else "slm::xtr32_mfma_kernel(slm::SplitArgs)")
slots = max(16, lanes) if lanes <= 20 else 32
counters = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_hbm_counters.json")))["kernels"][kernel]
old = json.load(open(os.path.join(ROOT, "profiles", "roofline_traffic.json")))
algorithmic = 8.0 * (n * p + slots * n + slots * p)
out = {
    "workload": {"n": n, "p": p, "lanes": lanes},
    "kernel": kernel,
    "hbm_bytes_per_launch": counters["hbm_bytes_per_launch"],
    "read_bytes_corrected": counters["read_bytes_corrected"],
    "write_bytes": counters["write_bytes"],
    "ratio_to_algorithmic_bytes": counters["hbm_bytes_per_launch"] / algorithmic,
    "working_launches": counters["FETCH_SIZE"]["working_launches"],
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1 --cpu-budget 0 "
              f"--no-extra` (profiles/{tag}_hbm_counters.json); KiB->bytes; FETCH_SIZE doubled per the gfx950 correction of "
              "MI355X_MICROARCH.md section HBM (16-byte-per-lane global loads); median over working launches",
    "source": f"profiles/{tag}_hbm_counters.json",
    "taken_on": {
        "commit": subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip(),
        "kernel_source": src,
        "kernel_source_sha256": hashlib.sha256(open(os.path.join(ROOT, src), "rb").read()).hexdigest(),
    },
    "other_kernels": old.get("other_kernels", {}),
}
json.dump(out, open(os.path.join(ROOT, "profiles", "roofline_traffic.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("hbm_bytes_per_launch", "ratio_to_algorithmic_bytes", "taken_on")}, indent=1))
