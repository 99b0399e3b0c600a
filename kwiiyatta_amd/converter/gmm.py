"""Innermost stage of the converter stack: a joint-density Gaussian mixture over [source | target] dynamic features.
API of kwiiyatta.converter.gmm (/root/reference/kwiiyatta/converter/gmm.py): the mixture object sits in `.gmm`,
`_train` fits it, `convert` runs maximum-likelihood parameter generation.  Both run on the GPU: the fit is
GaussianMixtureHIP (k-means + EM kernels, scikit-learn's GaussianMixture.fit semantics), the conversion the MLPG
kernels behind kwiiyatta_amd.backend.mlpg (nnmnkwii's MLPG call signature)."""
from ..backend.mlpg import MLPG
from . import abc, delta
from .gmm_fit import GaussianMixtureHIP as GaussianMixture


class GMMFeatureConverter(abc.FeatureConverter):
    def __init__(self, components=64, max_iter=100, random_state=None, **kwargs):
        self.init_gmm(components, max_iter, random_state, **kwargs)

    def init_gmm(self, components, max_iter=100, random_state=None, **kwargs):
        options = dict(verbose=1, covariance_type='full')
        options.update(kwargs)
        self.gmm = GaussianMixture(n_components=components, max_iter=max_iter, random_state=random_state, **options)

    def _train(self, dataarray, **kwargs):
        self.gmm.fit(dataarray, **kwargs)

    def convert(self, feature, mlpg=True, diff=False):
        # mlpg=False: the static window alone -- every frame converted on its own (posterior-weighted conditional mean)
        windows = delta.DELTA_WINDOWS if mlpg else delta.DELTA_WINDOWS[0:1]
        return MLPG(self.gmm, windows=windows, diff=diff).transform(feature)
