"""Import alias: the package directory is named `semantic-search_amd` (not a valid Python
identifier), so this module presents it as `semantic_search_amd`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "semantic-search_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
