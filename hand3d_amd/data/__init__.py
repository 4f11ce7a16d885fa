from .BinaryDbReader import BinaryDbReader, BinaryDbReaderSTB  # noqa: F401
