"""Built-in pattern strings (constraints/format.rs:237-294) for Python-side tools; the C++ host layer holds
its own copy (term_amd/csrc/host/term_guard.cpp FormatType::get_pattern)."""
EMAIL = (r"^[a-zA-Z0-9.!#$%&'*+/=?^_`{|}~-]+@[a-zA-Z0-9](?:[a-zA-Z0-9-]{0,61}[a-zA-Z0-9])?"
         r"(?:\.[a-zA-Z0-9](?:[a-zA-Z0-9-]{0,61}[a-zA-Z0-9])?)*$")
