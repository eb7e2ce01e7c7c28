"""Forward-output taps on sub-modules addressed by dotted path (the "Hookability" row of SURVEY.md section 8(b);
call sites libs/cil/cil.py:432-446 and :526-527, which construct ``OutputHook(model, names, as_tensor=True)`` and read
``get_layer_output(name)`` after a forward).

Contract kept from the reference's plugin surface: constructor arguments, ``get_layer_output``, ``remove``, use as a
context manager, ``AttributeError`` for a path that does not resolve, and taps that survive pickling (the reference
spawns one process per GPU and pickles the LightningModule with its hooks).  The tapped tensor is handed out as is, so
a feature-distillation loss back-propagates through it."""
from __future__ import annotations

from typing import Dict, Iterable, Optional


def rgetattr(obj, path: str, *default):
    """``getattr`` along a dotted path: ``rgetattr(model, 'backbone.layer1')``; an optional default applies at each step."""
    cur = obj
    for part in path.split('.'):
        cur = getattr(cur, part, *default)
    return cur


class _Tap:
    """The callable given to ``register_forward_hook``: a plain module-level class (picklable) that keeps the most recent
    output of the module it is attached to."""

    __slots__ = ('keep_tensor', 'output')

    def __init__(self, keep_tensor: bool):
        self.keep_tensor = keep_tensor
        self.output = None

    def __call__(self, module, inputs, output):
        self.output = output if self.keep_tensor else output.detach().cpu().numpy()

    def __getstate__(self):
        return {'keep_tensor': self.keep_tensor}            # the captured activation does not travel

    def __setstate__(self, state):
        self.keep_tensor = state['keep_tensor']
        self.output = None


class OutputHook:
    """Taps the forward output of each named sub-module of ``module``."""

    def __init__(self, module, outputs: Optional[Iterable[str]] = None, as_tensor: bool = True):
        self.outputs = outputs
        self.as_tensor = as_tensor
        self._taps: Dict[str, _Tap] = {}
        self._handles = []
        self.register(module)

    def register(self, module):
        names = self.outputs if isinstance(self.outputs, (list, tuple)) else ()
        resolved = []
        for name in names:
            try:
                resolved.append((name, rgetattr(module, name)))
            except AttributeError:
                raise AttributeError(f'Module {name} not found') from None
        for name, target in resolved:                       # attach only once every path has resolved
            tap = _Tap(self.as_tensor)
            self._handles.append(target.register_forward_hook(tap))
            self._taps[name] = tap

    def tap(self, layer_name: str) -> _Tap:
        return self._taps[layer_name]

    def get_layer_output(self, layer_name: str):
        return self._taps[layer_name].output

    def remove(self):
        while self._handles:
            self._handles.pop().remove()

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.remove()
