from typing import Optional, Tuple, Union
from torch import Tensor
OptTensor = Optional[Tensor]
PairTensor = Tuple[Tensor, Tensor]
OptPairTensor = Tuple[Tensor, Optional[Tensor]]
PairOptTensor = Tuple[Optional[Tensor], Optional[Tensor]]
Adj = Tensor
Size = Optional[Tuple[int, int]]
NoneType = type(None)
