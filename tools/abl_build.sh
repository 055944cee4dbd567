# build variant libraries of the current sources with extra -D flags: bash tools/abl_build.sh name "-DFLAG ..."
set -e
name=$1; shift
cd $(dirname $0)/..
mkdir -p tools/abl
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 "$@" -o tools/abl/$name.so guassianhand_amd/csrc/gh_api.hip guassianhand_amd/csrc/gh_preprocess.hip guassianhand_amd/csrc/gh_binning.hip guassianhand_amd/csrc/gh_render.hip guassianhand_amd/csrc/gh_uv.hip guassianhand_amd/csrc/gh_sh.hip guassianhand_amd/csrc/gh_knn.hip guassianhand_amd/csrc/gh_loss.hip guassianhand_amd/csrc/gh_select.hip 2>&1 | grep -E "error" || true
ls -la tools/abl/$name.so
