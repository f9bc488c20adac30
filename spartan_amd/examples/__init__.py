"""Workload drivers for the two application configs of BASELINE.json: `lreg` (configs[4], least squares
by gradient steps) and `sklearn.cluster.KMeans` (configs[3]).  They are thin driver loops over the
expression API; every per-tile body runs in HIP kernels through the backend."""
