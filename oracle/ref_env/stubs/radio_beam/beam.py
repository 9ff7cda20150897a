import numpy as np
SIGMA_TO_FWHM = np.sqrt(8 * np.log(2))
FWHM_TO_AREA = 2 * np.pi / (8 * np.log(2))


class NoBeamException(Exception):
    pass
