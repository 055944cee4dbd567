# GRBM cycles and VALU instruction counts of every launch of tools/micro/valu_cycles.hip: pins the issue cost of each instruction
# kind in cycles of the clock the SQ counters of profiles/*_pmc_sq.json are counted in (usage: bash tools/micro/valu_cycles_pmc.sh <out dir>)
O=${1:-gpurun_out/r3}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p $O
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_cycles tools/micro/valu_cycles.hip || exit 1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES -d $O/vc_pmc -o vc --output-format csv -- /tmp/valu_cycles > $O/valu_cycles_under_pmc.txt 2>&1 || exit 1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
f = glob.glob(O + "/vc_pmc/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# one entry per (KIND, grid): the LAST launch of each (the second of two)
import re
disp = collections.OrderedDict()
for r in rows:
    m = re.search(r"k<(\d+)>", r["Kernel_Name"])
    if not m:
        continue
    key = (int(m.group(1)), int(r["Grid_Size"]), int(r["Dispatch_Id"]))
    disp.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
last = {}
for (kind, grid, did), c in disp.items():
    if (kind, grid) not in last or did > last[(kind, grid)][0]:
        last[(kind, grid)] = (did, c)
names = {0: "v_fma_f32", 1: "v_mul_f32", 2: "v_add_f32", 21: "v_sub_f32", 4: "v_min_f32", 22: "v_max_f32", 3: "v_cndmask_b32 (sgpr mask)",
         7: "v_and_b32", 24: "v_bfe_u32", 8: "v_cmp_gt_f32", 9: "v_rndne_f32", 10: "v_ldexp_f32", 13: "v_cvt_i32_f32", 11: "v_rcp_f32",
         12: "v_exp_f32", 5: "v_mov_b32_dpp quad_perm", 6: "v_add_f32_dpp quad_perm", 20: "v_mul_f32_dpp quad bcast",
         17: "v_add_f32_dpp row_shr:1", 18: "s_nop 1 + v_add_dpp row_shr", 19: "v_add_f32_dpp row_bcast:15", 15: "v_pk_fma_f32",
         16: "v_pk_mul_f32"}
grids = {1: 256 * 256, 2: 256 * 512, 4: 256 * 1024, 8: 512 * 1024}
out = open(O + "/valu_cycles_pmc.txt", "w")
print("GRBM_GUI_ACTIVE / 8 (cycles of the clock the SQ counters count in) per wave64 VALU instruction per SIMD = cycles x 1024 SIMDs / "
      "SQ_INSTS_VALU,\nW waves resident per SIMD, 8 independent chains per lane; last column: the GRBM clock of the W = 8 launch "
      "(cycles / HIP-event time is not available under the profiler: see valu_cycles.txt for the wall-clock rates)", file=out)
for kind, n in names.items():
    line = f"{n:28s}"
    for W, grid in grids.items():
        ent = last.get((kind, grid))
        if not ent:
            line += f"  W={W}    n/a"
            continue
        c = ent[1]
        valu = c.get("SQ_INSTS_VALU", 0.0)
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0        # the counter is summed over the 8 XCDs
        line += f"  W={W} {cyc * 1024.0 / valu if valu else float('nan'):6.2f}"
    print(line, file=out)
out.close()
print(open(O + "/valu_cycles_pmc.txt").read())
PY
rm -rf $O/vc_pmc
