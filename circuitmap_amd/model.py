"""`Model` with the reference's interface for the CAVIaR path (reference circuitmap/model.py:16-44,104-162)."""
import time
from copy import deepcopy

import numpy as np

from . import optimise


class Model:
    def __init__(self, N, priors=None):
        """Initialise circuitmap model (defaults: reference circuitmap/model.py:24-31)."""
        self.N = N
        self.priors = priors if priors is not None else {}
        self.priors.setdefault('alpha', 1/4 * np.ones(N))
        self.priors.setdefault('phi', np.c_[1e-1 * np.ones(N), 5e0 * np.ones(N)])
        self.priors.setdefault('phi_cov', np.array([np.array([[1e-1, 0], [0, 1e0]]) for _ in range(N)]))
        self.priors.setdefault('mu', np.zeros(N))
        self.priors.setdefault('beta', 1e1 * np.ones(N))
        self.priors.setdefault('shape', 1.)
        self.priors.setdefault('rate', 1e-1)
        self.state = deepcopy(self.priors)

    def fit(self, obs, stimuli, method='caviar', fit_options=dict()):
        if method == 'caviar':
            self._fit_caviar(obs, stimuli, fit_options)
        elif method == 'cavi_sns':
            # the reference's own cavi_sns cannot run: its update_lam reads `key` before assigning it inside body_fun
            # (circuitmap/optimise/cavi_sns.py:138-146 unpacks the carry as `current_key`, then splits `key`) -> UnboundLocalError
            raise NotImplementedError("method 'cavi_sns' is not implemented: the reference's implementation raises UnboundLocalError "
                                      "(cavi_sns.py:146), so there is no behaviour to reproduce; use method='caviar'")
        elif method == 'mbcs':
            raise NotImplementedError("method 'mbcs' is outside the MI355X hot-path scope (SURVEY.md section 2); "
                                      "only method='caviar' is implemented")
        else:
            raise Exception          # reference circuitmap/model.py:44

    def _fit_caviar(self, obs, stimuli, fit_options):
        t_start = time.time()
        result = optimise.caviar(
            obs, stimuli, self.state['mu'], self.state['beta'], self.state['shape'],
            self.state['rate'], self.state['phi'], self.state['phi_cov'], **fit_options
        )
        t_end = time.time()
        names = ['mu', 'beta', 'lam', 'shape', 'rate', 'phi', 'phi_cov', 'z', 'receptive_fields']
        for name, val in zip(names, result[:9]):
            # the reference copies device arrays into fresh NumPy arrays (model.py:118-126); optimise.caviar already returns fresh
            # host arrays, so they are adopted without a second 16 MB copy of lam.  np.array(None) for receptive_fields, as :126.
            self.state[name] = np.asarray(val) if val is not None else np.array(val)
        self.trial_count = self.state['lam'].shape[1]
        self.time = t_end - t_start
        self.history = {k: (np.asarray(v) if v is not None else np.array(v)) for k, v in zip(['mu', 'beta', 'lam', 'shape', 'rate', 'phi', 'phi_cov', 'z'], result[9:17])}
        return
