"""Config objectification and the string->class plugin loader.

Host-side mirror of the reference's plugin boundary (src/utils/util.py:25-71 ``DictionaryUtility``,
:74-137 ``ModuleUtility``): JSON dictionaries become attribute objects, modules named in an
``imports`` list are imported and keyed by the LAST component of their dotted name, and class
strings such as ``"spconv.SubMConv3d"`` are resolved against those keys.  Listing
``"waveformml_amd.spconv"`` in ``imports`` therefore binds the key ``spconv`` to the MI355X
implementation without touching the rest of a reference config.
"""
import importlib
import json


class DictionaryUtility(object):
    @staticmethod
    def to_object(item):
        """dict -> attribute object, recursively (lists are converted element-wise)."""
        if isinstance(item, dict):
            return type("jo", (), {k: DictionaryUtility.to_object(v) for k, v in item.items()})
        if isinstance(item, list):
            return [DictionaryUtility.to_object(v) for v in item]
        return item

    @staticmethod
    def to_dict(obj):
        """attribute object -> dict, recursively; names starting with '_' are dropped."""
        if not hasattr(obj, "__dict__"):
            return obj
        out = {}
        for key, val in obj.__dict__.items():
            if key.startswith("_"):
                continue
            if isinstance(val, list):
                out[key] = [DictionaryUtility.to_dict(v) for v in val]
            else:
                out[key] = DictionaryUtility.to_dict(val)
        return out


class ModuleUtility(object):
    """Imports the listed modules and instantiates classes named by "<module key>.<Class>" strings."""

    def __init__(self, modlist):
        self.modList = modlist
        self.modules = {}
        self.classes = {}
        for dotted in modlist:
            self.modules[dotted.split(".")[-1]] = importlib.import_module(dotted)

    def retrieve_module(self, name):
        if name not in self.modules:
            raise IOError("{0} is not in the module list.".format(name))
        return self.modules[name]

    def retrieve_class(self, class_string):
        if class_string in self.classes:
            return self.classes[class_string]
        if "." in class_string:
            parts = class_string.split(".")
            cls = getattr(self.retrieve_module(parts[0]), parts[1])
        else:
            for mod in self.modules.values():
                if hasattr(mod, class_string):
                    cls = getattr(mod, class_string)
                    break
            else:
                raise IOError("{0} is not a valid class path.\n"
                              " Must be formatted <module name>.<class name>".format(class_string))
        self.classes[class_string] = cls
        return cls

    def create_class_instances(self, classes):
        """``["mod.Cls", [args], "mod.Other", [args], ...]`` -> instances.  A class string that is
        not followed by an argument list contributes the CLASS itself (reference util.py:114-115,135-136)."""
        instances = []
        pending = None                      # class string waiting for its argument list
        for entry in classes:
            if isinstance(entry, str):
                if pending is not None:
                    instances.append(self.retrieve_class(pending))
                pending = entry
            elif isinstance(entry, list):
                if pending is None:
                    raise IOError("Argument list must be preceded by a string of the class path.\n"
                                  "Errored at input: ", str(entry))
                instances.append(self.retrieve_class(pending)(*entry))
                pending = None
            elif isinstance(entry, dict):
                for key in entry:
                    instances.append(self.retrieve_class(entry[key])(**classes[entry[key]]))
                pending = None
        if pending is not None:
            instances.append(self.retrieve_class(pending))
        return instances


def load_config(path_or_dict):
    """JSON file (or dict) -> attribute object, as reference main.py:87-95 does."""
    if isinstance(path_or_dict, dict):
        return DictionaryUtility.to_object(path_or_dict)
    with open(path_or_dict) as f:
        return DictionaryUtility.to_object(json.load(f))
