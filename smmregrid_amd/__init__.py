"""smmregrid_amd -- MI355X-native sparse regrid engine (drop-in for the
apply path of jhardenberg/smmregrid: Regridder(...).regrid(), CdoGenerate)."""
from .regrid import Regridder, regrid
from .cdogenerate import CdoGenerate, cdo_generate_weights
from .gridtype import GridType
from .gridmeta import CdoGrid, GridDetector, GridInspector
from .operator import SparseOperator, OperatorGroup
from .device import DeviceArray, pinned_empty, to_device
from .xrlite import DataArray, Dataset

__version__ = '0.1.0'

__all__ = ["Regridder", "regrid", "CdoGenerate", "cdo_generate_weights", "GridType", "GridInspector", "GridDetector", "CdoGrid",
           "SparseOperator", "OperatorGroup", "DeviceArray", "to_device", "pinned_empty", "DataArray", "Dataset"]
