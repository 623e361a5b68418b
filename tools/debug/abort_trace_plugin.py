"""pytest -p tools.debug.abort_trace_plugin: builds tools/debug/abort_trace.c next to itself and installs its SIGABRT handler
(the native stack of whoever called abort(), which Python's faulthandler cannot show).  Run with --capture=sys: under pytest's
default fd capture a runtime library's own last words on stderr (a GPU memory fault's address, a queue error) go to the
capture file and are lost with the process — which is why such an abort looks silent."""
import ctypes
import os
import subprocess

_here = os.path.dirname(os.path.abspath(__file__))
_so = os.path.join(_here, "_abort_trace.so")
subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", "-rdynamic", os.path.join(_here, "abort_trace.c"), "-o", _so])
_lib = ctypes.CDLL(_so)
_lib.ipcfp_debug_abort_trace_keep_stderr()  # (this module is imported before pytest's capture starts)


def pytest_sessionstart(session):
    _lib.ipcfp_debug_install_abort_trace()
