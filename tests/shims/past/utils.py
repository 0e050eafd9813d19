import numbers


def old_div(a, b):
    """Python-2 division: floor for two integers, true division otherwise."""
    if isinstance(a, numbers.Integral) and isinstance(b, numbers.Integral):
        return a // b
    return a / b
