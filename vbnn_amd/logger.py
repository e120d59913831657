"""The run logger of logger.lua, file for file: one text file per series id under the run directory, one value per
line -- exactly what visualize.py:25-39 (`read_data` / `get_data`: float(line.strip()) per line) reads back.

    Log = Logger(opt.network_name, append=False)      # logger.lua:5-12  (main.lua:148,151)
    Log.add("devacc", 97.3)                           # logger.lua:18-26
    Log.flush(); Log.close()                          # logger.lua:36-46

Semantics kept: `init` creates the directory if it is missing (`mkdir`, logger.lua:6); the first `add` of an id
truncates the file unless the logger was opened in append mode (logger.lua:20-22), later adds append; values are
written with Lua's `..` number formatting (`%.14g`, so 97.3 -> "97.3", 1e-05 -> "1e-05") followed by a newline.
`append(id, value)` (logger.lua:28-34) reopens in append mode and adds. Host-only; no device code.
"""
import os


def lua_tostring(value):
    """Lua 5.1 `tostring(number)` / `..` on a number: "%.14g"; strings pass through."""
    if isinstance(value, str):
        return value
    if isinstance(value, bool):
        raise TypeError("logger.lua concatenates numbers or strings; a boolean raises in Lua too")
    return "%.14g" % float(value)


class Logger:
    def __init__(self, directory, append=False):                  # logger.lua:5-12
        os.makedirs(directory, exist_ok=True)
        self.dir = directory
        self.loggers = {}
        self.append_mode = bool(append)

    def _create(self, series, mode):                              # logger.lua:14-16
        self.loggers[series] = open(os.path.join(self.dir, series), mode)

    def add(self, series, value):                                 # logger.lua:18-26
        if series not in self.loggers:
            if not self.append_mode:
                open(os.path.join(self.dir, series), "w").close()
            self._create(series, "a")
        self.loggers[series].write(lua_tostring(value) + "\n")

    def append(self, series, value):                              # logger.lua:28-34
        if series not in self.loggers or not self.append_mode:
            if series in self.loggers:
                self.loggers[series].close()
            self._create(series, "a+")
            self.append_mode = True
        self.add(series, value)

    def flush(self):                                              # logger.lua:36-40
        for f in self.loggers.values():
            f.flush()

    def close(self):                                              # logger.lua:42-46
        for f in self.loggers.values():
            f.close()
        self.loggers = {}


def read_data(filename):
    """visualize.py:25-31, restated for tests: the floats of a series file, [] if any line is not a number."""
    with open(filename) as f:
        try:
            return [float(x.strip()) for x in f.readlines()]
        except ValueError:
            return []
