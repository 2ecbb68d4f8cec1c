"""pytracer_amd -- the MI355X path behind pytracer's ``ImageTracer.fire_all_rays``.

Importing the package has no side effects.  One process-wide knob is the CALLER's to turn:

``prefer_device_kernargs()`` asks the HIP runtime to keep kernel ARGUMENTS in device memory (``HIP_FORCE_DEV_KERNARG=1``,
unless the variable is already set): by default the runtime places the kernarg segment in host memory, so the first
scalar load of every wave of a launch crosses the host link (~1 us).  Measured on the MI355X
(profiles/r04_dev_kernarg.txt): C2 14.5 -> 13.5 us, OnOff 13.8 -> 11.2 us, C5 56 -> 50 us per frame.  The runtime reads
the variable when it initialises, i.e. the call must precede the first HIP call of the process (also torch's), and it
changes HIP for every user of the runtime in the process -- which is why neither the package nor ``libptrace.so`` does it
behind the caller's back (round 4 did).  The ``render`` command and ``bench.py`` call it first thing;
``pytracer_amd.device.device_kernargs()`` reports what is in effect.
"""
import os as _os


def prefer_device_kernargs() -> bool:
    """Set ``HIP_FORCE_DEV_KERNARG=1`` unless the variable is set already; -> True iff device kernargs are now asked
    for.  Only effective before the process's first HIP call."""
    _os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    return _os.environ.get("HIP_FORCE_DEV_KERNARG", "0") not in ("", "0")
