from .axial_vit import *
from ._api import *
