# FETCH_SIZE calibration for this pipeline's access shapes (usage on the GPU box: bash tools/micro/fetch_calib.sh <out dir>)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r5}
hipcc --offload-arch=gfx950 -O2 -o /tmp/fetch_calib tools/micro/fetch_calib.hip 2>/dev/null || exit 1
/tmp/fetch_calib > $O/fetch_calib.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fc -o fc --output-format csv -- /tmp/fetch_calib > /dev/null 2>&1
python3 tools/summarize_pmc.py $(find $O/fc -name "*counter_collection.csv" | head -1) >> $O/fetch_calib.txt
rm -rf $O/fc
