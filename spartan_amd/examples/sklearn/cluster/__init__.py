from .k_means_ import KMeans  # noqa: F401
