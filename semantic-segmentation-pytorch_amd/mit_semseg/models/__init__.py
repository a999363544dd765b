from .models import ModelBuilder, SegmentationModule
