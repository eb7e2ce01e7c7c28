"""Picklable forward-hook taps addressed by dotted module name (libs/module_hooks/output_hook.py:16-69)."""
import functools


class OutputHookWrapper:
    def __init__(self, as_tensor=True):
        self.as_tensor = as_tensor
        self.output = None

    def __call__(self, module, input, output) -> None:
        self.output = output if self.as_tensor else output.detach().cpu().numpy()


class OutputHook:
    def __init__(self, module, outputs=None, as_tensor=True):
        self.outputs = outputs
        self.as_tensor = as_tensor
        self._layer_outputs = {}
        self.handles = []
        self.register(module)

    def register(self, module):
        if isinstance(self.outputs, (list, tuple)):
            for name in self.outputs:
                try:
                    layer = rgetattr(module, name)
                except AttributeError:
                    raise AttributeError(f'Module {name} not found')
                hook = OutputHookWrapper(as_tensor=self.as_tensor)
                self.handles.append(layer.register_forward_hook(hook))
                self._layer_outputs[name] = hook

    def remove(self):
        for h in self.handles:
            h.remove()

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.remove()

    def get_layer_output(self, layer_name):
        return self._layer_outputs[layer_name].output


def rgetattr(obj, attr, *args):
    def _getattr(o, a):
        return getattr(o, a, *args)
    return functools.reduce(_getattr, [obj] + attr.split('.'))
