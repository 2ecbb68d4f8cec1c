"""pytracer_amd -- the MI355X path behind pytracer's ``ImageTracer.fire_all_rays``.

Importing the package asks the HIP runtime to keep kernel ARGUMENTS in device memory (``HIP_FORCE_DEV_KERNARG=1``, unless
the caller has set the variable): by default the runtime places the kernarg segment in host memory, so the first scalar
load of every wave of a launch crosses the host link (~1.5 us) -- a tenth of a 14-us frame.  Measured on the MI355X
(profiles/r04_dev_kernarg.txt): C2 14.5 -> 13.5 us, OnOff 13.8 -> 11.2 us, C5 56 -> 50 us per frame.  The variable is read
when the runtime initialises, i.e. it must be set before the first HIP call of the process: import this package (or
``libptrace.so``, whose load-time constructor does the same) before touching the GPU.
"""
import os as _os

_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
