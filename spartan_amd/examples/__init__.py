"""Driver programs of the reference's examples that BASELINE.json benchmarks
(spartan/examples/): the SGD regressions (configs[4], benchmark_lreg.py) and
k-means (configs[3], benchmark_kmeans.py / tests/test_kmeans.py).  Same class
and function names and argument meaning as the reference; every per-tile body
runs in HIP kernels through the backend."""
