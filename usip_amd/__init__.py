"""usip_amd -- MI355X (gfx950) implementation of the USIP detector hot path.

Hand-written HIP kernels behind a C ABI (include/usip_hip.h, libusip_hip.so), exposed through
the reference's own Python surface:

  usip_amd.dropin.index_max / ball_query   the two extension modules networks.py imports
  usip_amd.layers / losses / operations / som   nn.Modules and helpers with the reference's
                                                class names, constructors and state_dict keys
  usip_amd.networks                        RPN_Detector / RPN_Detector_Ball built on the above
  usip_amd.step                            ModelDetector.optimize as one data-parallel step

`usip_amd.install()` registers the drop-in extension modules under their reference names
(`import index_max`, `import ball_query`) so that the reference's models/networks.py runs
unchanged on top of this package.  There is no CPU fallback anywhere in the product path.
"""
__version__ = "0.1"


def install():
    """Make `import index_max` / `import ball_query` resolve to the HIP implementations."""
    import sys
    from .dropin import ball_query, index_max
    sys.modules["index_max"] = index_max
    sys.modules["ball_query"] = ball_query
    return index_max, ball_query
