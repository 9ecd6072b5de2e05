#!/usr/bin/env python3
"""Summary of a rocprofv3 PC-sampling CSV (stochastic or host-trap) for the kernels named in KERNELS: per kernel the samples by stall reason / instruction type,
and the instructions that collect the most samples, each with its own stall-reason split.
    python tools/pcsamp_summary.py <pc_sampling csv> [<kernel_trace csv>]
The column names differ between rocprofiler-sdk versions: they are looked up case-insensitively by substring."""
import collections
import csv
import sys

KERNELS = ("lg_blockf_mixed_kernel", "attention32_kernel", "lg_blockf_kernel")


def col(fields, *subs):
    for f in fields:
        if all(s in f.lower() for s in subs):
            return f
    return None


def main(path, ktrace=None):
    disp = {}
    if ktrace:
        with open(ktrace) as fh:
            rd = csv.DictReader(fh)
            kd, kn = col(rd.fieldnames, "dispatch"), col(rd.fieldnames, "kernel_name") or col(rd.fieldnames, "name")
            for r in rd:
                disp[r[kd]] = r[kn]
    with open(path) as fh:
        rd = csv.DictReader(fh)
        f = rd.fieldnames
        print("columns:", f)
        c_inst = col(f, "instruction") if col(f, "instruction") and "comment" not in col(f, "instruction").lower() else None
        c_inst = next((x for x in f if x.lower() == "instruction"), c_inst)
        c_disp = col(f, "dispatch")
        c_stall = col(f, "stall")
        c_type = col(f, "instruction_type") or col(f, "inst_type")
        c_issued = col(f, "issued")
        c_cmt = col(f, "comment")
        per = collections.defaultdict(lambda: dict(n=0, stall=collections.Counter(), typ=collections.Counter(), issued=0, inst=collections.defaultdict(lambda: [0, collections.Counter(), ""])))
        for r in rd:
            name = disp.get(r.get(c_disp, ""), "") if disp else ""
            key = next((k for k in KERNELS if k in name), None) if disp else "all"
            if key is None:
                continue
            if disp:
                key = name.split("(")[0].replace("void airfe::", "")
            p = per[key]
            p["n"] += 1
            st = r.get(c_stall, "") if c_stall else ""
            p["stall"][st] += 1
            if c_type:
                p["typ"][r[c_type]] += 1
            if c_issued and r[c_issued] not in ("0", "", "false", "False"):
                p["issued"] += 1
            i = p["inst"][r.get(c_inst, "?")]
            i[0] += 1
            i[1][st] += 1
            if c_cmt and not i[2]:
                i[2] = r[c_cmt]
    for k, p in sorted(per.items(), key=lambda kv: -kv[1]["n"]):
        print(f"\n== {k}: {p['n']} samples, issued {p['issued']} ({p['issued'] / max(p['n'], 1):.1%})")
        print("  by stall reason:", ", ".join(f"{s or '-'} {n / p['n']:.1%}" for s, n in p["stall"].most_common()))
        if p["typ"]:
            print("  by instruction type:", ", ".join(f"{s or '-'} {n / p['n']:.1%}" for s, n in p["typ"].most_common()))
        print("  top instructions:")
        for ins, (n, st, cmt) in sorted(p["inst"].items(), key=lambda kv: -kv[1][0])[:40]:
            print(f"    {n / p['n']:6.2%}  {ins[:70]:70s} {', '.join(f'{s or chr(45)} {m}' for s, m in st.most_common(3))}  {cmt[-60:]}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
