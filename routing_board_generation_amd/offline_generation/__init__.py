from .dataset_generator import BoardDatasetGenerator  # noqa: F401
