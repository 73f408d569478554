"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from .binding import OracleAsgram, OracleChain, OracleDsd, OrcCfg, OrcTaps, build, firdes_kaiser, lib  # noqa: F401
