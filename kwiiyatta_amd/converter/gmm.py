"""Joint-density GMM converter (mirrors
/root/reference/kwiiyatta/converter/gmm.py:9-34).  Apply = MLPG kernels, fit = EM
kernels (GaussianMixtureHIP reproduces sklearn.mixture.GaussianMixture.fit for the
configuration the reference uses)."""
from .gmm_fit import GaussianMixtureHIP as GaussianMixture

from ..backend.mlpg import MLPG
from . import abc, delta


class GMMFeatureConverter(abc.FeatureConverter):
    def __init__(self, components=64, max_iter=100, random_state=None, **kwargs):
        super().__init__()
        self.init_gmm(components, max_iter, random_state, **kwargs)

    def init_gmm(self, components, max_iter=100, random_state=None, **kwargs):
        kwargs.setdefault('verbose', 1)
        kwargs.setdefault('covariance_type', 'full')
        self.gmm = GaussianMixture(n_components=components, max_iter=max_iter,
                                   random_state=random_state, **kwargs)

    def _train(self, dataarray, **kwargs):
        self.gmm.fit(dataarray, **kwargs)

    def convert(self, feature, mlpg=True, diff=False):
        if not mlpg:
            raise NotImplementedError('frame-wise (mlpg=False) conversion is not used by the '
                                      'reference CLIs and not implemented on the GPU')
        return MLPG(self.gmm, windows=delta.DELTA_WINDOWS, diff=diff).transform(feature)
