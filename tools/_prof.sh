set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04prof
python tools/lreg_phases.py > gpurun_out/r04prof/lreg_phases.txt 2>&1
ROWS=70 python tools/driver_profile.py lreg > gpurun_out/r04prof/lreg_cum.txt 2>&1
SORT=tottime ROWS=60 python tools/driver_profile.py lreg > gpurun_out/r04prof/lreg_tot.txt 2>&1
ROWS=70 python tools/driver_profile.py kmeans > gpurun_out/r04prof/kmeans_cum.txt 2>&1
SORT=tottime ROWS=60 python tools/driver_profile.py kmeans > gpurun_out/r04prof/kmeans_tot.txt 2>&1
cat gpurun_out/r04prof/lreg_phases.txt
