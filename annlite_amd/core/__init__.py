from .codec import PQCodec
