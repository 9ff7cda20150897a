class BeamError(Exception):
    pass


class NoBeamException(Exception):
    pass


class RadioBeamDeprecationWarning(Warning):
    pass
