# kernel boundary cost inside a replayed graph, by kernel shape (usage: bash tools/micro/kernel_floor.sh <out dir>)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r5}
hipcc --offload-arch=gfx950 -O2 -o /tmp/kernel_floor tools/micro/kernel_floor.hip 2>/dev/null || exit 1
/tmp/kernel_floor > $O/kernel_floor.txt
rocprofv3 --kernel-trace --stats -d $O/kf -o kf --output-format csv -- /tmp/kernel_floor > /dev/null 2>&1
python3 tools/micro/kernel_floor_stats.py $O/kf >> $O/kernel_floor.txt
rm -rf $O/kf
