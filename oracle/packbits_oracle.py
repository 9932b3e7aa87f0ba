"""CPU restatement of the reference's PackBits (src/codec/packbits.py) -- TEST INFRASTRUCTURE ONLY (tests/ import it as the
checker of csrc/packbits_kernels.hip; the product never does).

A byte-at-a-time state machine, each step citing the reference lines it follows, written as functions of a fresh state
(one call = one fresh PackBits object of the reference).  Parity pinned by the known answers the reference's own
`__main__` demo and SURVEY Appendix C hold (tests/test_packbits_oracle.py).
"""

MAX_LENGTH = 127  # packbits.py:30


def delta_transform(data):
    """packbits.py:43-51"""
    out = [data[0]]
    for i in range(1, len(data)):
        out.append((data[i] - data[i - 1]) % 256)
    return out


def revert_delta_transform(data):
    """packbits.py:53-63"""
    out = [data[0]]
    for i in range(1, len(data)):
        d = data[i] - 256 if data[i] > 127 else data[i]
        out.append((out[i - 1] + d) % 256)
    return out


def encode(data, apply_delta_transform=False):
    """packbits.py:74-129"""
    if len(data) == 0:
        return bytearray()
    if len(data) == 1:
        return bytearray(b"\x00" + bytes(bytearray(data)))
    if apply_delta_transform:
        data = delta_transform(data)
    data = bytearray(data)
    result, buf = bytearray(), bytearray()
    state, run, pos = "RAW", 0, 0
    while pos < len(data) - 1:  # packbits.py:91
        if data[pos] == data[pos + 1]:
            if state == "RAW":  # packbits.py:95-99: flush the literals, start a run
                if buf:
                    result.append(len(buf) - 1); result.extend(buf); buf = bytearray()
                state, run = "RLE", 1
            else:  # packbits.py:101-107
                if run == MAX_LENGTH:
                    result.append(256 - (run - 1)); result.append(data[pos]); run = 0
                run += 1
        else:
            if state == "RLE":  # packbits.py:111-115: the run ends with this byte
                run += 1
                result.append(256 - (run - 1)); result.append(data[pos])
                state, run = "RAW", 0
            else:  # packbits.py:117-121
                if len(buf) == MAX_LENGTH:
                    result.append(len(buf) - 1); result.extend(buf); buf = bytearray()
                buf.append(data[pos])
        pos += 1
    if state == "RAW":  # packbits.py:125-127: the last byte joins the open literals unchecked
        buf.append(data[pos])
        result.append(len(buf) - 1); result.extend(buf)
    else:  # packbits.py:128-130
        run += 1
        result.append(256 - (run - 1)); result.append(data[pos])
    return result


def decode(data, apply_delta_transform=False):
    """packbits.py:131-163"""
    data = bytearray(data)
    result, pos = bytearray(), 0
    while pos < len(data):
        header = data[pos] - 256 if data[pos] > 127 else data[pos]
        pos += 1
        if 0 <= header <= 127:
            result.extend(data[pos: pos + header + 1])
            pos += header + 1
        elif header == -128:
            pass
        else:
            result.extend([data[pos]] * (1 - header))
            pos += 1
    if apply_delta_transform and len(result) > 0:
        return revert_delta_transform(result)
    return result
