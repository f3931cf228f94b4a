"""Detector geometry and sidereal time for the gravitational-wave leg -- the host-side constants the C ABI takes
(``nmma_gw_config.detector_tensor / vertex / gmst_ref``).

The reference gets these from bilby (``bilby.gw.detector``: ``Interferometer.vertex``, ``.detector_tensor``,
``bilby_cython.time.greenwich_mean_sidereal_time``); bilby is absent from the build image, so the same published
definitions are restated here: WGS-84 vertex position and arm unit vectors from the site description
(``bilby/gw/detector/geometry.py``, ``detectors/{H1,L1,V1}.interferometer``) and LAL's GMST polynomial
(``XLALGreenwichMeanSiderealTime``).  Real bilby ``Interferometer`` objects carry the same attribute names and are accepted
wherever an :class:`Interferometer` of this module is.
"""
from __future__ import annotations

import math

import numpy as np

#: latitude [deg], longitude [deg], elevation [m], x-arm azimuth [deg], y-arm azimuth [deg], x-arm tilt, y-arm tilt [rad]
SITES = {
    "H1": (46 + 27. / 60 + 18.528 / 3600, -(119 + 24. / 60 + 27.5657 / 3600), 142.554, 125.9994, 215.9994, -6.195e-4, 1.25e-5),
    "L1": (30 + 33. / 60 + 46.4196 / 3600, -(90 + 46. / 60 + 27.2654 / 3600), -6.574, 197.7165, 287.7165, -3.121e-4, -6.107e-4),
    "V1": (43 + 37. / 60 + 53.0921 / 3600, 10 + 30. / 60 + 16.1887 / 3600, 51.884, 70.5674, 160.5674, 0.0, 0.0),
}
#: GPS seconds at which a leap second took effect (bilby_cython.time.LEAP_SECONDS)
LEAP_SECONDS_GPS = (46828800, 78364801, 109900802, 173059203, 252028804, 315187205, 346723206, 393984007, 425520008,
                    457056009, 504489610, 551750411, 599184012, 820108813, 914803214, 1025136015, 1119744016, 1167264017)
_SEMI_MAJOR, _SEMI_MINOR = 6378137.0, 6356752.314


def site_geometry(latitude_deg, longitude_deg, elevation, xarm_azimuth_deg, yarm_azimuth_deg, xarm_tilt=0.0, yarm_tilt=0.0):
    """``(vertex[3] metres, detector_tensor[3, 3])`` of an L-shaped interferometer."""
    lat, lon = math.radians(latitude_deg), math.radians(longitude_deg)
    radius = _SEMI_MAJOR ** 2 / math.sqrt(_SEMI_MAJOR ** 2 * math.cos(lat) ** 2 + _SEMI_MINOR ** 2 * math.sin(lat) ** 2)
    vertex = np.array([(radius + elevation) * math.cos(lat) * math.cos(lon),
                       (radius + elevation) * math.cos(lat) * math.sin(lon),
                       ((_SEMI_MINOR / _SEMI_MAJOR) ** 2 * radius + elevation) * math.sin(lat)])
    e_long = np.array([-math.sin(lon), math.cos(lon), 0.0])
    e_lat = np.array([-math.sin(lat) * math.cos(lon), -math.sin(lat) * math.sin(lon), math.cos(lat)])
    e_h = np.array([math.cos(lat) * math.cos(lon), math.cos(lat) * math.sin(lon), math.sin(lat)])

    def arm(tilt, azimuth_deg):
        az = math.radians(azimuth_deg)
        return math.cos(tilt) * math.cos(az) * e_long + math.cos(tilt) * math.sin(az) * e_lat + math.sin(tilt) * e_h

    x, y = arm(xarm_tilt, xarm_azimuth_deg), arm(yarm_tilt, yarm_azimuth_deg)
    return vertex, 0.5 * (np.outer(x, x) - np.outer(y, y))


def greenwich_mean_sidereal_time(gps_time):
    """GMST in radians (not wrapped) at a GPS time: UTC Julian day from the leap-second table, then the IAU polynomial
    in Julian centuries since J2000.0, integer and fractional seconds kept apart as LAL does."""
    leaps = sum(1 for g in LEAP_SECONDS_GPS if gps_time >= g)
    julian_day = 2444244.5 + (math.floor(gps_time) - leaps) / 86400.0      # GPS epoch = 1980-01-06T00:00:00 UTC
    t_hi = (julian_day - 2451545.0) / 36525.0
    t_lo = (gps_time % 1.0) / (36525.0 * 86400.0)
    t = t_hi + t_lo
    s = (-6.2e-6 * t + 0.093104) * t * t + 67310.54841
    s += 8640184.812866 * t_lo
    s += 3155760000.0 * t_lo
    s += 8640184.812866 * t_hi
    s += 3155760000.0 * t_hi
    return s * math.pi / 43200.0


def gmst_linearisation(reference_gps_time, half_width=64.0):
    """``(gmst_ref, gmst_rate)`` with gmst(t) ~ gmst_ref + gmst_rate (t - reference): the quadratic term of the polynomial
    changes the rate by 1e-19 rad/s per second, far below fp64 resolution over any coalescence-time prior."""
    a = greenwich_mean_sidereal_time(reference_gps_time - half_width)
    b = greenwich_mean_sidereal_time(reference_gps_time + half_width)
    return greenwich_mean_sidereal_time(reference_gps_time), (b - a) / (2.0 * half_width)


class _StrainData:
    def __init__(self, start_time, duration, sampling_frequency):
        self.start_time, self.duration, self.sampling_frequency = float(start_time), float(duration), float(sampling_frequency)


class Interferometer:
    """The attributes of ``bilby.gw.detector.Interferometer`` the likelihood reads: ``name``, ``frequency_array``,
    ``frequency_domain_strain``, ``power_spectral_density_array``, ``frequency_mask`` (from ``minimum_frequency`` /
    ``maximum_frequency``), ``strain_data.start_time`` / ``.duration``, ``vertex``, ``detector_tensor``, ``time_array``."""

    def __init__(self, name, frequency_domain_strain, power_spectral_density_array, duration, start_time,
                 minimum_frequency=20.0, maximum_frequency=None, sampling_frequency=None, geometry=None):
        self.name = name
        self.frequency_domain_strain = np.asarray(frequency_domain_strain, dtype=np.complex128)
        self.power_spectral_density_array = np.asarray(power_spectral_density_array, dtype=np.float64)
        n = self.frequency_domain_strain.shape[0]
        if self.power_spectral_density_array.shape != (n,):
            raise ValueError("strain and PSD must have the same length")
        self.frequency_array = np.arange(n) / float(duration)
        if sampling_frequency is None:
            sampling_frequency = 2.0 * (n - 1) / float(duration)
        self.strain_data = _StrainData(start_time, duration, sampling_frequency)
        self.minimum_frequency = float(minimum_frequency)
        self.maximum_frequency = float(sampling_frequency / 2.0 if maximum_frequency is None else maximum_frequency)
        self.vertex, self.detector_tensor = site_geometry(*SITES[name]) if geometry is None else geometry

    @property
    def frequency_mask(self):
        f = self.frequency_array
        return (f >= self.minimum_frequency) & (f <= self.maximum_frequency)

    @property
    def time_array(self):
        n = int(round(self.strain_data.duration * self.strain_data.sampling_frequency))
        return self.strain_data.start_time + np.arange(n) / self.strain_data.sampling_frequency

    @property
    def duration(self):
        return self.strain_data.duration

    @property
    def start_time(self):
        return self.strain_data.start_time
