from .grid import GridEncoder, grid_encode
