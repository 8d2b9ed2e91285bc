from .dataset_generator import BoardDatasetGenerator, BoardDatasetGeneratorJAX  # noqa: F401
