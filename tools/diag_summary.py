"""profiles/rNN_parity_diag_summary.txt from gpurun_out/diag/*.json (tests/gpu_common.py::diag): one line per record, lists dropped.
    rm -rf gpurun_out/diag; <run the -m gpu suite on the MI355X through gpurun>; python tools/diag_summary.py profiles/r04_parity_diag_summary.txt "<header>" """
import glob
import json
import os
import sys

out, header = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for f in sorted(glob.glob(os.path.join("gpurun_out", "diag", "*.json"))):
    d = json.load(open(f))
    vals = []
    for k, v in d.items():
        if isinstance(v, list):
            continue
        vals.append(f"{k}={round(v, 4) if isinstance(v, float) else v}")
    rows.append(f"{os.path.basename(f)[:-5]:44s} " + "  ".join(vals))
with open(out, "w") as fh:
    fh.write("# Numbers the -m gpu parity tests recorded on the MI355X (tests/gpu_common.py::diag)\n# " + header + "\n\n" + "\n".join(rows) + "\n")
print(f"{len(rows)} records -> {out}")
