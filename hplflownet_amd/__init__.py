"""hplflownet_amd -- MI355X-native bilateral-convolution hot path of HPLFlowNet.

Public surface (mirrors the reference's operator interface for the path):
    BilateralConvFlex, BilateralCorrelationFlex, sparse_sum     (layers; bcl.py)
    GenerateDataUnsymmetric                                      (GPU lattice; lattice.py)
    HPLFlowNet, HPLFlowNetShallow, DeviceLattice                 (callers; flownet.py)
The arithmetic lives in libhplbcl.so (csrc/*.hip, C ABI in include/hpl_bcl.h).
"""
from .bcl import (BilateralConvFlex, BilateralCorrelationFlex, Conv1dReLU, Conv2dReLU, Conv3dReLU,  # noqa: F401
                  sparse_sum)
from .flownet import DeviceLattice, HPLFlowNet, HPLFlowNetShallow  # noqa: F401
from .lattice import GenerateDataUnsymmetric, to_reference_format  # noqa: F401

__version__ = '0.1.0'
