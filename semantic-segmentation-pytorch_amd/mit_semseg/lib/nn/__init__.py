from .modules import *
from .parallel import UserScatteredDataParallel, user_scattered_collate, async_copy_to
