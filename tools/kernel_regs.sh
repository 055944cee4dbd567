# Register / LDS / scratch use of every kernel in one source file (usage: bash tools/kernel_regs.sh gh_render.hip)
F=${1:-gh_render.hip}
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 --offload-device-only -S -o /tmp/kregs.s guassianhand_amd/csrc/$F || exit 1
python3 - <<'PY'
import re
t = open('/tmp/kregs.s').read()
for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)', t):
    pass
blocks = t.split('- .agpr_count:')[1:] if '- .agpr_count:' in t else []
for b in blocks:
    g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, b) or [None, '?'])[1]
    print(f"{g('name')[:70]:70s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>4s}")
PY
