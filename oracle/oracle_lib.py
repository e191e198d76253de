"""ORACLE loader (test infrastructure only): ctypes handle on oracle/librda_oracle.so."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librda_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("rda_oracle.c", "lmz_ipm.c", "rda_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def load():
    build()
    return C.CDLL(_SO)
