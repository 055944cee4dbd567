# build tools/abl/<name>.so from the sources of a COMMIT (default HEAD): the "before" arm of a same-box A/B
# usage: bash tools/abl_base.sh [commit] [name]
set -e
C=${1:-HEAD}; NAME=${2:-base}
cd $(dirname $0)/..
rm -rf /tmp/gh_base && mkdir -p /tmp/gh_base tools/abl
git archive $C guassianhand_amd/csrc include | tar -x -C /tmp/gh_base
S=/tmp/gh_base/guassianhand_amd/csrc
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 -o tools/abl/$NAME.so $S/gh_api.hip $S/gh_preprocess.hip $S/gh_binning.hip $S/gh_render.hip $S/gh_uv.hip $S/gh_sh.hip $S/gh_knn.hip $S/gh_loss.hip $S/gh_select.hip
ls -la tools/abl/$NAME.so
