"""Pairs tools/tile_shape_sweep.py's launch log with a rocprofv3 kernel trace: python tile_shape_post.py log.json kernel_trace.csv"""
import csv, json, sys
log = json.load(open(sys.argv[1]))
rows = [r for r in csv.DictReader(open(sys.argv[2])) if "k_resize" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
i = 0
cur = None
for g in log:
    ks = rows[i:i + g["launches"]]
    i += g["launches"]
    if not ks:
        break
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in ks)
    if g["case"] != cur:
        cur = g["case"]
        print(f"\n{cur}")
    name = ks[0]["Kernel_Name"].split("(")[0].replace("void vpf::", "")
    shape = "policy     " if g["ty"] == 0 else f"ty {g['ty']:2d} wpb {g['wpb']}"
    grid = int(ks[0].get("Grid_Size_X", ks[0].get("Grid_Size", "0")) or 0)
    print(f"  {shape}: median {d[len(d) // 2]:7.2f} us  min {d[0]:7.2f}  {'' if g['same_pixels'] else 'PIXELS DIFFER '} {name} lds {ks[0].get('LDS_Block_Size', '?')} wg {ks[0].get('Workgroup_Size_X', '?')}")
print(f"\n{i} of {len(rows)} traced resize kernels consumed")
