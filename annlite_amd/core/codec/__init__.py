from .pq import DistanceTable, PQCodec
