"""Which trained segmentation model to load (reference: robotpose/training/models.py:19-324, used by
Predictor.__init__ as `ModelManager().dynamicLoad(dataset=model_ds)`, predict.py:94-98).

Every trained model is a folder `<MODELS>/<id>/` holding `ModelData.json` (id, dataset, sizes, classes, date trained)
and its Keras checkpoints `*.<epoch>-<loss>.h5`.  This is the read side only: the reference's manager also prunes old
checkpoints, deletes empty folders and rewrites an index file whenever it is constructed; selecting a model for
prediction has no business deleting anything, so none of that is done here."""
import json
import logging
import os
from datetime import datetime

import numpy as np

from .config import Paths

MODELDATA_FILE_NAME = 'ModelData.json'      # constants.py:37

_FIELDS = {'id': '', 'dataset': '', 'dataset_size': 0, 'train_size': 0, 'valid_size': 0, 'classes': [], 'epochs_trained': 0,
           'date_trained': '', 'benchmarks': []}
_STATIC = {'dataset', 'classes', 'benchmark'}
_DYNAMIC = {'dataset_size', 'train_size', 'valid_size', 'train_ratio', 'valid_ratio', 'used_ratio', 'epochs_trained'}


class ModelData:
    """One trained model's record (models.py:19-53), plus the ratios the selection rules may name."""

    def __init__(self, source=None, **kwargs):
        self.__dict__.update({k: (list(v) if isinstance(v, list) else v) for k, v in _FIELDS.items()})
        if isinstance(source, str):
            path = source if source.endswith(MODELDATA_FILE_NAME) else os.path.join(source, MODELDATA_FILE_NAME)
            with open(path) as f:
                source = json.load(f)
        for d in (source or {}, kwargs):
            self.__dict__.update({k: v for k, v in d.items() if k in _FIELDS})
        n = self.dataset_size or 1
        self.train_ratio, self.valid_ratio = self.train_size / n, self.valid_size / n
        self.used_ratio = (self.train_size + self.valid_size) / n

    def __getitem__(self, key):
        return self.__dict__[key]

    def __repr__(self):
        return str({k: self.__dict__[k] for k in _FIELDS})


def _epoch_of(checkpoint: str) -> int:
    # mask_rcnn_model.<epoch>-<val_loss>.h5 (models.py:89)
    try:
        return int(checkpoint.split('.')[1].split('-')[0])
    except (IndexError, ValueError):
        return 0


class ModelManager:

    def __init__(self, models_dir: str = None):
        self.dir = models_dir or Paths().MODELS
        self.update()

    def update(self):
        self.info = {}
        if not os.path.isdir(self.dir):
            return
        for root, _, files in os.walk(self.dir):
            if MODELDATA_FILE_NAME in files:
                data = ModelData(os.path.join(root, MODELDATA_FILE_NAME))
                data.folder = root
                data.epochs_trained = max([_epoch_of(x) for x in files if x.endswith('.h5')] + [0])
                self.info[data.id] = data
        self.num_total = len(self.info)

    def loadByID(self, id: str) -> str:
        """The last checkpoint (by name, i.e. by epoch) of a model (models.py:180-190)."""
        assert id in self.info, f"id {id} not found"
        folder = self.info[id].folder
        files = sorted(f for f in os.listdir(folder) if f.endswith('.h5'))
        if not files:
            raise FileNotFoundError(f"model {id} has no checkpoint in {folder}")
        return os.path.join(folder, files[-1])

    def dynamicLoad(self, kwarg_dict: dict = None, **kwargs):
        """Path of the checkpoint of the 'best' model under the given criteria, applied in the order given
        (models.py:192-324).  Static criteria (dataset, classes) filter when anything satisfies them and are dropped
        with a warning otherwise; dynamic ones (dataset_size, train_size, valid_size, *_ratio, epochs_trained) pick the
        closest value, +-inf the extreme, and `<name>_above` / `<name>_below` filter, falling back to the extreme.
        Several survivors: the most recently trained.  None when no model exists."""
        if kwarg_dict is not None:
            kwargs.update(kwarg_dict)
        known = _STATIC | _DYNAMIC | {k + s for k in _DYNAMIC for s in ('_above', '_below')}
        for key in kwargs:
            assert key in known, f"Unknown kwarg '{key}'"
        remaining = self._apply(dict(self.info), kwargs)
        if not remaining:
            return None
        if len(remaining) > 1:
            logging.info(f"SEG MODEL SELECTION: {len(remaining)} models match the chosen selection. Choosing most recently trained.")
        newest = max(remaining.values(), key=lambda m: datetime.strptime(m.date_trained, '%Y-%m-%d %H:%M:%S.%f'))
        return self.loadByID(newest.id)

    @staticmethod
    def _apply(remaining: dict, criteria: dict) -> dict:
        def extreme(models, key, fn):
            best = fn(getattr(m, key) for m in models.values())
            return {k: m for k, m in models.items() if getattr(m, key) == best}

        for key, value in criteria.items():
            if len(remaining) <= 1:
                return remaining
            before = remaining
            if key in _STATIC:
                if key == 'benchmark':
                    continue                                   # a TODO in the reference as well
                remaining = {k: m for k, m in remaining.items() if getattr(m, key) == value}
                if not remaining:
                    remaining = before
                    logging.warning(f"Not using {key}={value} for model selection; Not satisfied by any remaining models.")
            elif key.endswith('_above') or key.endswith('_below'):
                above = key.endswith('_above')
                attr = key.rsplit('_', 1)[0]                   # the reference reads the suffixed name itself (an AttributeError)
                remaining = {k: m for k, m in remaining.items() if (getattr(m, attr) >= value if above else getattr(m, attr) <= value)}
                if not remaining:
                    logging.warning(f"{key}={value} not satisfied for model selection; Using {'maximum' if above else 'minimum'} value instead.")
                    return extreme(before, attr, max if above else min)
            elif abs(value) == np.inf:
                return extreme(remaining, key, max if value > 0 else min)
            else:
                closest = min(abs(value - getattr(m, key)) for m in remaining.values())
                remaining = {k: m for k, m in remaining.items() if abs(value - getattr(m, key)) == closest}
        return remaining
