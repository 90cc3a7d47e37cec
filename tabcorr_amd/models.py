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


LEAUTHAUD11_KEYS = ('smhm_m0_0', 'smhm_m0_a', 'smhm_m1_0', 'smhm_m1_a',
                    'smhm_beta_0', 'smhm_beta_a', 'smhm_delta_0',
                    'smhm_delta_a', 'smhm_gamma_0', 'smhm_gamma_a',
                    'scatter_model_param1', 'alphasat', 'bsat', 'betasat',
                    'bcut', 'betacut')


def behroozi10_log_halo_mass(log_stellar_mass, log_m0, log_m1, beta, delta,
                             gamma, littleh):
    """Behroozi, Conroy & Wechsler (2010), eq. 21: log10 of the halo mass (in
    1 / h units) of galaxies of stellar mass ``10**log_stellar_mass`` (in
    1 / h^2 units)."""
    x = log_stellar_mass + 2.0 * np.log10(littleh) - log_m0
    return (log_m1 + beta * x + 10.0**(delta * x) / (1.0 + 10.0**(-gamma * x))
            - 0.5 - np.log10(littleh))


def behroozi10_log_stellar_mass(log_halo_mass, log_m0, log_m1, beta, delta,
                                gamma, littleh):
    """Inverse of `behroozi10_log_halo_mass`, solved to rounding by Newton's
    method from the right of the root."""
    log_halo_mass = np.asarray(log_halo_mass, dtype=np.float64)
    target = log_halo_mass + np.log10(littleh) + 0.5 - log_m1
    x = target / beta
    with np.errstate(all='ignore'):
        alt = np.log10(np.maximum(2.0 * target, 1e-300)) / delta
        x = np.where(target > 0.5, np.minimum(x, alt), x)
        for _ in range(60):
            a = 10.0**(delta * x)
            b = 10.0**(-gamma * x)
            g = beta * x + a / (1.0 + b)
            slope = beta + np.log(10.0) * a / (1.0 + b) * (
                delta + gamma * b / (1.0 + b))
            step = (g - target) / slope
            x = x - step
            if not np.any(np.abs(step) > 1e-15 * np.maximum(1.0, np.abs(x))):
                break
    return x + log_m0 - 2.0 * np.log10(littleh)


class Leauthaud11Model:
    """Leauthaud et al. (2011) HOD on the Behroozi et al. (2010)
    stellar-to-halo mass relation, with halotools' parameter names
    (``PrebuiltHodModelFactory('leauthaud11')``: ``smhm_*``,
    ``scatter_model_param1``, ``alphasat``, ``bsat``, ``betasat``, ``bcut``,
    ``betacut``) and defaults.

    The occupation functions are restated from the papers with halotools'
    conventions (h = 0.7 inside the Behroozi et al. relation, h = 0.72 in the
    satellite terms, knee mass 1e12, scatter sqrt(2) sigma); unlike halotools,
    which inverts the stellar-to-halo mass relation by spline interpolation of
    a 100-point table, the inverse is solved exactly -- the two agree to the
    accuracy of that table, not to rounding, which is why only THIS class
    (not a halotools ``Leauthaud11Cens`` / ``Leauthaud11Sats`` composite) is
    evaluated by the device kernel.

    Parameters
    ----------
    threshold : float
        log10 of the stellar mass threshold of the sample.
    redshift : float
    modulate_with_cenocc : bool
        Multiply <N_sat> by <N_cen> (halotools' default for this family).
    """

    _tabcorr_amd_device_model = 'leauthaud11'
    # halotools' composite carries two Hubble parameters: the stellar-to-halo
    # mass relation (Behroozi10SmHm) was calibrated with h = 0.7, the satellite
    # terms of Leauthaud11Sats use h = 0.72
    littleh_smhm = 0.7
    littleh_sats = 0.72

    def __init__(self, threshold=10.5, prim_haloprop_key='halo_mvir',
                 redshift=0.0, modulate_with_cenocc=True, **param_dict):
        self.gal_types = ['centrals', 'satellites']
        self.threshold = threshold
        self.redshift = redshift
        self.modulate_with_cenocc = modulate_with_cenocc
        self._input_model_dictionary = {
            'centrals_occupation': _OccupationComponent(prim_haloprop_key),
            'satellites_occupation': _OccupationComponent(prim_haloprop_key)}
        self.param_dict = {
            'smhm_m0_0': 10.72, 'smhm_m0_a': 0.59, 'smhm_m1_0': 12.35,
            'smhm_m1_a': 0.3, 'smhm_beta_0': 0.43, 'smhm_beta_a': 0.18,
            'smhm_delta_0': 0.56, 'smhm_delta_a': 0.18, 'smhm_gamma_0': 1.54,
            'smhm_gamma_a': 2.52, 'scatter_model_param1': 0.2,
            'alphasat': 1.0, 'bsat': 10.62, 'betasat': 0.859, 'bcut': 1.47,
            'betacut': -0.13}
        self.param_dict.update(param_dict)

    def device_theta(self):
        """The 14 columns the kernel reads (``include/tabcorr_amd.h``,
        TC_FLAG_LEAUTHAUD11): the relation at the model's redshift,
        a = 1 / (1 + z), X = X_0 + X_a (a - 1), ..., the Hubble parameter of
        the relation and that of the satellite terms."""
        p = self.param_dict
        a = 1.0 / (1.0 + self.redshift)
        smhm = [p['smhm_%s_0' % k] + p['smhm_%s_a' % k] * (a - 1.0)
                for k in ('m0', 'm1', 'beta', 'delta', 'gamma')]
        return np.array(smhm + [p['scatter_model_param1'], p['alphasat'],
                                p['bsat'], p['betasat'], p['bcut'],
                                p['betacut'], self.threshold,
                                self.littleh_smhm, self.littleh_sats],
                        dtype=np.float64)

    def mean_occupation_centrals(self, prim_haloprop=None, **kwargs):
        return leauthaud11_centrals(prim_haloprop, self.device_theta())

    def mean_occupation_satellites(self, prim_haloprop=None, **kwargs):
        return leauthaud11_satellites(prim_haloprop, self.device_theta(),
                                      self.modulate_with_cenocc)


def leauthaud11_centrals(prim_haloprop, theta):
    """Leauthaud et al. (2011), eq. 8."""
    log_mstar = behroozi10_log_stellar_mass(
        np.log10(prim_haloprop), theta[0], theta[1], theta[2], theta[3],
        theta[4], theta[12])
    return 0.5 * (1.0 - _erf((theta[11] - log_mstar) /
                             (np.sqrt(2.0) * theta[5])))


def leauthaud11_satellites(prim_haloprop, theta, modulate_with_cenocc=True):
    """Leauthaud et al. (2011), eq. 12 (knee mass 1e12)."""
    prim_haloprop = np.asarray(prim_haloprop, dtype=np.float64)
    littleh = theta[13]
    knee = 10.0**behroozi10_log_halo_mass(
        theta[11], theta[0], theta[1], theta[2], theta[3], theta[4],
        theta[12]) * littleh
    m_sat = 1e12 * theta[7] * (knee / 1e12)**theta[8]
    m_cut = 1e12 * theta[9] * (knee / 1e12)**theta[10]
    n = (np.exp(-m_cut / (prim_haloprop * littleh)) *
         (prim_haloprop * littleh / m_sat)**theta[6])
    if modulate_with_cenocc:
        n = n * leauthaud11_centrals(prim_haloprop, theta)
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

    def __init__(self, theta, modulate_with_cenocc=False, assembias=False,
                 family='zheng07'):
        # (a list of floats: the un-batched path copies it straight into its
        # staging array; `theta` makes the array the batched paths want)
        self.values = theta
        self.modulate_with_cenocc = bool(modulate_with_cenocc)
        self.assembias = bool(assembias)
        self.family = family


DeviceSpec.theta = property(
    lambda self: np.asarray(self.values, dtype=np.float64))


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

    if getattr(model, '_tabcorr_amd_device_model', None) == 'leauthaud11':
        cls = type(model)
        if (getattr(cls, 'mean_occupation_centrals', None) is not
                Leauthaud11Model.mean_occupation_centrals or
                getattr(cls, 'mean_occupation_satellites', None) is not
                Leauthaud11Model.mean_occupation_satellites or
                getattr(cls, 'device_theta', None) is not
                Leauthaud11Model.device_theta or
                'mean_occupation_centrals' in vars(model) or
                'mean_occupation_satellites' in vars(model)):
            return None
        try:
            theta = model.device_theta()
        except KeyError:
            return None
        return DeviceSpec(theta, model.modulate_with_cenocc, False,
                          family='leauthaud11')

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
