from .dataset import Dataset
