import warnings

MUTE = True


def info(msg, *args):
    pass


def debug(msg, *args):
    pass


def warn(msg, *args):
    if not MUTE:
        warnings.warn(msg % args if args else msg)


def error(msg, *args):
    print('gym-shim ERROR:', msg % args if args else msg)
