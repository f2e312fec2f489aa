"""Mode and filter-bank label tables the Receiver is constructed with
(reference ``Tables.py:34-62``; passed at ``receiver.py:835``)."""

MODES = ["AM", "AM-Synch", "SSB", "USB", "LSB", 'CW', "IQ", "WFM", "WFM2", "NFM", "RTTY"]

AF_BWs = ['Max', '50 Hz', '100 Hz', '500 Hz', '1 KHz', '2 KHz', '3 KHz',
          '4 KHz', '5 KHz', '8 KHz', '10 KHz', '15 KHz', '20 KHz', '45 KHz', '50 KHz',
          '100 KHz', '150 KHz', '200 KHz']

VIDEO_BWs = ['Max', '5 KHz', '10 KHz', '20 KHz', '25 KHz', '45 KHz', '50 KHz', '100 KHz',
             '150 KHz', '200 KHz', '300 KHz', '400 KHz', '500 KHz', '750 KHz', '1 MHz', 'Other']

RTLsrates = [0.25, 1.024, 1.536, 1.792, 1.92, 2.048, 2.16, 2.56, 2.88, 3.2]
SDRplaysrates = [0.25, 0.5, 1, 2, 2.048, 3, 4, 5, 6, 7, 8, 9, 10]

MODE_INDEX = {m: i for i, m in enumerate(MODES)}
MODE_INDEX['FM'] = MODE_INDEX['NFM']          # receiver.py:639-640


def label_hz(label):
    """'5 KHz' -> 5000.0, '1 MHz' -> 1e6, 'Max'/'Other' -> None."""
    if label in ('Max', 'Other'):
        return None
    num, unit = label.split(' ')
    scale = {'Hz': 1.0, 'KHz': 1e3, 'MHz': 1e6}[unit]
    return float(int(num)) * scale


def find_filter(max_bw, bw_list):
    """Widest labelled filter not exceeding ``max_bw`` (``Tables.py:48-62``)."""
    best = None
    for bw in bw_list:
        hz = label_hz(bw)
        if hz is not None and hz <= max_bw:
            best = bw
    return best


def index_of_bw(bw_hz, labels, default):
    for i, lab in enumerate(labels):
        if label_hz(lab) == bw_hz:
            return i
    return default
