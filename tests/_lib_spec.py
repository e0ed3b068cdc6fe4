from term_amd._lib import spec  # noqa: F401
