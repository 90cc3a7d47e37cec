"""HOD models the device evaluates itself, and how foreign models are mapped.

The reference asks a halotools ``HodModelFactory`` for
``mean_occupation_centrals`` / ``mean_occupation_satellites`` at every
Gauss-Legendre node (``tabcorr/tabcorr.py:556-563``).  For the Zheng et al.
(2007) family those two functions are evaluated inside the HIP occupation
kernel; this module defines

* `Zheng07Model`, a light stand-in for
  ``PrebuiltHodModelFactory('zheng07', ...)`` exposing the attributes
  ``TabCorr.predict`` reads (``gal_types``, ``param_dict``, ``redshift``,
  ``_input_model_dictionary``) so that MCMC code written against halotools
  runs unchanged without halotools installed, and
* `device_spec`, which recognises models (this class, or halotools objects
  built from ``Zheng07Cens`` / ``Zheng07Sats`` and their assembly-bias
  decorated variants) whose occupation the kernel can evaluate and extracts
  the parameter vector.  Anything else takes the generic route: callbacks on
  the host, contraction on the device.
"""

import math

import numpy as np

ZHENG07_KEYS = ('logMmin', 'sigma_logM', 'logM0', 'logM1', 'alpha')
ASSEMBIAS_KEYS = ('mean_occupation_centrals_assembias_param1',
                  'mean_occupation_satellites_assembias_param1')

try:
    from scipy.special import erf as _erf
except ImportError:  # pragma: no cover
    _erf = np.vectorize(math.erf, otypes=[np.float64])


class _OccupationComponent:
    """Mimics the two attributes of a halotools occupation component that the
    consistency checks look at (``tabcorr/tabcorr.py:506-526``)."""

    def __init__(self, prim_haloprop_key, sec_haloprop_key=None):
        self.prim_haloprop_key = prim_haloprop_key
        if sec_haloprop_key is not None:
            self.sec_haloprop_key = sec_haloprop_key


class Zheng07Model:
    """Zheng et al. (2007) HOD, optionally with Heaviside assembly bias.

    Parameters
    ----------
    prim_haloprop_key : str, optional
        Halo mass definition the model is a function of.
    redshift : float, optional
    modulate_with_cenocc : bool, optional
        Multiply <N_sat> by <N_cen> (option of halotools' ``Zheng07Sats``).
    sec_haloprop_key : str or None, optional
        If given the model carries Heaviside assembly bias in this secondary
        property (strengths ``mean_occupation_*_assembias_param1`` in
        ``param_dict``, split at the median).
    **param_dict
        Initial parameter values; defaults are halotools' threshold -20
        ``zheng07`` values.
    """

    _tabcorr_amd_device_model = 'zheng07'

    def __init__(self, prim_haloprop_key='halo_mvir', redshift=0.0,
                 modulate_with_cenocc=False, sec_haloprop_key=None,
                 **param_dict):
        self.gal_types = ['centrals', 'satellites']
        self.redshift = redshift
        self.modulate_with_cenocc = modulate_with_cenocc
        self.assembias = sec_haloprop_key is not None
        self.split = 0.5
        self._input_model_dictionary = {
            'centrals_occupation': _OccupationComponent(
                prim_haloprop_key, sec_haloprop_key),
            'satellites_occupation': _OccupationComponent(
                prim_haloprop_key, sec_haloprop_key)}
        self.param_dict = {'logMmin': 12.02, 'sigma_logM': 0.26,
                           'logM0': 11.38, 'logM1': 13.31, 'alpha': 1.06}
        if self.assembias:
            for key in ASSEMBIAS_KEYS:
                self.param_dict[key] = 0.0
        self.param_dict.update(param_dict)

    # The two callbacks keep the model usable with the reference itself and
    # with the generic host route; the device path never calls them.
    def mean_occupation_centrals(self, prim_haloprop=None,
                                 sec_haloprop_percentile=None, **kwargs):
        p = self.param_dict
        n = 0.5 * (1.0 + _erf((np.log10(prim_haloprop) - p['logMmin']) /
                              p['sigma_logM']))
        if self.assembias:
            n = _heaviside(n, sec_haloprop_percentile, p[ASSEMBIAS_KEYS[0]],
                           1.0, self.split)
        return n

    def mean_occupation_satellites(self, prim_haloprop=None,
                                   sec_haloprop_percentile=None, **kwargs):
        p = self.param_dict
        prim_haloprop = np.asarray(prim_haloprop, dtype=np.float64)
        x = (prim_haloprop - 10.0**p['logM0']) / 10.0**p['logM1']
        n = np.where(x > 0, np.abs(x)**p['alpha'], 0.0)
        if self.modulate_with_cenocc:
            n = n * 0.5 * (1.0 + _erf(
                (np.log10(prim_haloprop) - p['logMmin']) / p['sigma_logM']))
        if self.assembias:
            n = _heaviside(n, sec_haloprop_percentile, p[ASSEMBIAS_KEYS[1]],
                           np.inf, self.split)
        return n


def _heaviside(baseline, percentile, strength, upper, split):
    f1, f2 = 1.0 - split, split
    # halotools clips the strength to [-1, 1] (NaN stays NaN)
    strength = 1.0 if strength > 1 else strength
    strength = -1.0 if strength < -1 else strength
    if strength >= 0:
        dmax = np.minimum(upper - baseline, baseline * f2 / f1)
    else:
        dmax = np.minimum(baseline, (upper - baseline) * f2 / f1)
    d1 = strength * dmax
    return np.where(np.asarray(percentile) > split, baseline + d1,
                    baseline - d1 * f1 / f2)


class DeviceSpec:
    """What the occupation kernel needs to know about one model."""

    def __init__(self, theta, modulate_with_cenocc=False, assembias=False):
        self.theta = np.asarray(theta, dtype=np.float64)
        self.modulate_with_cenocc = bool(modulate_with_cenocc)
        self.assembias = bool(assembias)


def device_spec(model):
    """Return a `DeviceSpec` if the kernel can evaluate ``model``, else None.

    The test is strict: the kernel implements exactly the plain Zheng07
    centrals / satellites and their Heaviside assembly-bias decoration with
    one constant strength per galaxy type and a split at the median.  A model
    that merely *looks* like one of those (same class names, but several
    strength abscissae, another split, an overridden occupation method, a
    satellite model modulated by something other than Zheng07 centrals)
    takes the generic route -- callbacks on the host, contraction on the
    device -- which is always correct.
    """
    if getattr(model, '_tabcorr_amd_device_model', None) == 'zheng07':
        cls = type(model)
        if (getattr(cls, 'mean_occupation_centrals', None) is not
                Zheng07Model.mean_occupation_centrals or
                getattr(cls, 'mean_occupation_satellites', None) is not
                Zheng07Model.mean_occupation_satellites or
                'mean_occupation_centrals' in vars(model) or
                'mean_occupation_satellites' in vars(model) or
                getattr(model, 'split', 0.5) != 0.5):
            return None
        return _spec_from(model, model.modulate_with_cenocc, model.assembias)

    components = getattr(model, '_input_model_dictionary', None)
    if not isinstance(components, dict):
        return None
    try:
        cens = components['centrals_occupation']
        sats = components['satellites_occupation']
    except KeyError:
        return None
    if _halotools_name(cens) == 'Zheng07Cens' and \
            _halotools_name(sats) == 'Zheng07Sats':
        assembias = False
    elif _halotools_name(cens) == 'AssembiasZheng07Cens' and \
            _halotools_name(sats) == 'AssembiasZheng07Sats':
        assembias = True
        for component in (cens, sats):
            if not _plain_heaviside(component):
                return None
    else:
        return None
    for component in (cens, sats):
        # an instance-level override of the occupation function
        if 'mean_occupation' in vars(component):
            return None
    modulate = bool(getattr(sats, 'modulate_with_cenocc', False))
    if modulate:
        # <N_sat> is multiplied by the occupation of
        # ``sats.central_occupation_model``: only plain Zheng07 centrals
        # (sharing logMmin / sigma_logM through the composite param_dict)
        # are what the kernel multiplies with
        if _halotools_name(getattr(sats, 'central_occupation_model',
                                   None)) != 'Zheng07Cens':
            return None
    param_dict = getattr(model, 'param_dict', {})
    allowed = set(ASSEMBIAS_KEYS) if assembias else set()
    for key in param_dict:
        if 'assembias' in str(key) and key not in allowed:
            return None      # e.g. ..._assembias_param2: mass-dependent strength
    return _spec_from(model, modulate, assembias)


def _halotools_name(component):
    """Class name of a halotools component, or None for anything that is not
    an instance of a class defined by halotools itself (a user subclass may
    override what the kernel assumes)."""
    cls = type(component)
    if not str(getattr(cls, '__module__', '')).startswith('halotools.'):
        return None
    return cls.__name__


def _plain_heaviside(component):
    """True if a HeavisideAssembias-decorated component has the only shape
    the kernel implements: one strength abscissa (constant strength) and a
    constant split at the median."""
    try:
        if len(component._assembias_strength_abscissa) != 1:
            return False
        ordinates = np.atleast_1d(np.asarray(
            component._split_ordinates, dtype=np.float64))
    except (AttributeError, TypeError, ValueError):
        return False
    return len(ordinates) >= 1 and bool(np.all(ordinates == 0.5))


def _spec_from(model, modulate, assembias):
    keys = ZHENG07_KEYS + (ASSEMBIAS_KEYS if assembias else ())
    try:
        theta = [float(model.param_dict[key]) for key in keys]
    except KeyError:
        return None
    return DeviceSpec(theta, modulate, assembias)
