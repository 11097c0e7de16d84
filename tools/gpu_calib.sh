# FETCH_SIZE / WRITE_SIZE calibration on a known byte count in libcrx's access pattern (8 B per lane, one wave per record)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/calib
mkdir -p $O
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/fetch_calib $R/tools/ubench/fetch_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
run() {  # tag counter args...
  tag=$1; ctr=$2; shift 2
  rm -rf $O/${tag}_$ctr
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${tag}_$ctr -o p -- /tmp/fetch_calib "$@" > $O/${tag}_$ctr.log 2>&1
}
# far beyond the 256 MiB Infinity Cache: 1 Mi records x 157 doubles = 1.3 GB read; x 118 doubles = 0.99 GB written
run big_read FETCH_SIZE read8 1048576 157 3
run big_read WRITE_SIZE read8 1048576 157 3
run big_write WRITE_SIZE write8 1048576 118 3
run big_write FETCH_SIZE write8 1048576 118 3
# the solver's own sizes: cfg2 (256 records: 39 doubles in, 118 out), cfg3 (4096: 57 in, 93 out)
run cfg2_read FETCH_SIZE read8 256 39 5
run cfg2_write WRITE_SIZE write8 256 118 5
run cfg3_read FETCH_SIZE read8 4096 57 5
run cfg3_write WRITE_SIZE write8 4096 93 5
python3 $R/profiles/summarize_calib.py
