"""SURVEY 8f #3 (text part): vbnn_amd/logger.py writes what logger.lua writes and visualize.py reads."""
import os

from vbnn_amd.logger import Logger, lua_tostring, read_data


def test_number_formatting_is_luas():
    assert lua_tostring(97.3) == "97.3"
    assert lua_tostring(3) == "3"
    assert lua_tostring(1e-5) == "1e-05"
    assert lua_tostring(0.1 + 0.2) == "0.3"              # %.14g, as Lua 5.1 prints it
    assert lua_tostring(123456789012345.0) == "1.2345678901234e+14"
    assert lua_tostring("LC") == "LC"


def test_truncate_then_append_and_read_back(tmp_path):
    d = os.path.join(tmp_path, "run")
    log = Logger(d)                                        # fresh run: first add truncates (logger.lua:20-22)
    for v in (0.5, 0.25, 1e-3):
        log.add("deverr", v)
    log.add("devacc", 91)
    log.flush(); log.close()
    assert read_data(os.path.join(d, "deverr")) == [0.5, 0.25, 1e-3]
    assert open(os.path.join(d, "devacc")).read() == "91\n"
    log = Logger(d)                                        # a new non-append logger starts the series over
    log.add("deverr", 7)
    log.close()
    assert read_data(os.path.join(d, "deverr")) == [7.0]
    log = Logger(d, append=True)                           # main.lua:148: resumed run keeps the history
    log.add("deverr", 8)
    log.append("devacc", 92)
    log.close()
    assert read_data(os.path.join(d, "deverr")) == [7.0, 8.0]
    assert read_data(os.path.join(d, "devacc")) == [91.0, 92.0]


def test_non_numeric_line_reads_as_empty(tmp_path):
    d = os.path.join(tmp_path, "run")
    log = Logger(d)
    log.add("notes", "nan-ish text")
    log.close()
    assert read_data(os.path.join(d, "notes")) == []       # visualize.py:28-30
