"""The slice of dimod the reference touches (/root/reference/src/losses.py:32,
/root/reference/src/utils/persistent_qpu_sampler.py:6,84-88): ``Sampler`` (annotation only), ``SampleSet``
(``.record.sample``, ``.record.energy``, ``.variables``, ``.vartype``, ``SampleSet.from_samples``), ``as_samples``."""
import numpy as np

from image_generation_amd.sampler import SampleSet as _SampleSet


class Sampler:
    """Base-class name used in type annotations."""


class SampleSet(_SampleSet):
    @classmethod
    def from_samples(cls, samples_like, energy=None, vartype="SPIN", **_unused):
        samples, labels = as_samples(samples_like)
        return cls(np.asarray(samples, dtype=np.int8), labels, energy=energy, vartype=vartype)


def as_samples(samples_like):
    """(2-D int8 array, variable labels): arrays get labels 0..n-1; (array, labels) pairs pass through."""
    if isinstance(samples_like, tuple) and len(samples_like) == 2:
        arr, labels = samples_like
        return np.atleast_2d(np.asarray(arr, dtype=np.int8)), list(labels)
    arr = np.atleast_2d(np.asarray(samples_like, dtype=np.int8))
    return arr, list(range(arr.shape[1]))
