from .base import BaseIndex
from .pq_flat_gpu import PQFlatGpuIndex
