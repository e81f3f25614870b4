"""Measured HBM traffic against algorithmic bytes, per workload and direction, from the committed summaries:
    python scripts/traffic_report.py <tag> [bench-tag]   reads profiles/<tag>_pmc_<workload>.json (+ profiles/<bench-tag or tag>_bench.json for the sizes)
Traffic = sum over the workload's kernels of FETCH_SIZE (doubled, gfx950) + WRITE_SIZE at each kernel's LARGEST launch x the number of
full-size launches a call makes of it (1, except kernels launched once per scan level, whose bytes are negligible).  Algorithmic bytes per
SURVEY 8d: Huffman encode 2N + C, decode C + N_out; LZSS encode N + C, decode C + N; config 4 = the sum of its two layers."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEC = ("k_lzd_", "k_une_", "k_dec_")


def main():
    tag = sys.argv[1]
    btag = sys.argv[2] if len(sys.argv) > 2 else tag
    bench = json.loads(open(os.path.join(ROOT, "profiles", btag + "_bench.json")).read().strip().splitlines()[-1])
    oc = bench.get("other_configs", {})
    n = 1 << 30
    rows = []
    for label, key in (("headline", None), ("skewed", "skewed"), ("config3", "3"), ("config4", "4")):
        p = os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (tag, label))
        if not os.path.exists(p):
            continue
        d = json.load(open(p))
        enc = sum(v.get("hbm_max", 0) for k, v in d.items() if not k.startswith(DEC))
        dec = sum(v.get("hbm_max", 0) for k, v in d.items() if k.startswith(DEC))
        if key is None:
            C = round(n * bench["ratio_pct"] / 100.0)
            alg_e, alg_d = 2 * n + C, C + n
        else:
            e = oc[key]
            C = round(n * e["ratio_pct"] / 100.0)
            if key == "4":
                l1 = e["layer_sizes"][0]
                alg_e, alg_d = (n + l1) + (2 * l1 + C), (C + l1) + (l1 + n)
            elif key == "3":
                alg_e, alg_d = n + C, C + n
            else:
                alg_e, alg_d = 2 * n + C, C + e["decoded_bytes"]
        rows.append((label, enc, alg_e, dec, alg_d))
    print("%-10s %14s %14s %7s   %14s %14s %7s" % ("workload", "encode traffic", "algorithmic", "ratio", "decode traffic", "algorithmic", "ratio"))
    for label, enc, ae, dec, ad in rows:
        print("%-10s %14.3e %14.3e %7.2f   %14.3e %14.3e %7.2f" % (label, enc, ae, enc / ae, dec, ad, dec / ad))
    print("\n(FETCH_SIZE is doubled for every kernel -- the gfx950 correction the guide gives for wide coalesced streaming reads, 16 B per lane.  Kernels whose"
          "\n loads are scattered -- k_chain_serial: one 16-byte group of keys per lane and step -- have no calibration: their true fetch lies between the raw"
          "\n counter (half the figure below) and the doubled one.)")
    print("\nlargest kernels per workload (fetch + write at the largest launch, GB):")
    for label in ("headline", "skewed", "config3", "config4"):
        p = os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (tag, label))
        if os.path.exists(p):
            d = json.load(open(p))
            top = sorted(d.items(), key=lambda kv: -kv[1].get("hbm_max", 0))[:6]
            print("  %-9s " % label + ", ".join("%s %.2f" % (k.split("<")[0], v.get("hbm_max", 0) / 1e9) for k, v in top))


if __name__ == "__main__":
    main()
