"""TEST INFRASTRUCTURE: restatement of the reference's NUT muxer for ONE raw video stream in bit-exact mode, so that the
`md5` the FATE suite takes over `-f nut md5:` (tests/fate-run.sh:458-466 video_filter, :468-497 pixfmts) can be reproduced
from frame bytes alone.  Follows /root/reference/ffmpeg-gpu/libavformat/nutenc.c:

  write order      nut_write_header :719-806 -> write_headers :664-717 (main, stream, global info, stream info);
                   nut_write_packet :960-1187 (a syncpoint before every frame: raw frames exceed max_distance);
                   nut_write_trailer :1189-1211 -> write_index :605-662
  frame code table build_frame_code :167-308 (one video stream, no delay), written by write_mainheader :369-452
  packets          put_packet :351-367: startcode, forward pointer, header checksum beyond 4096 bytes, payload, CRC
  checksum         ff_crc04C11DB7_update: CRC-32 polynomial 0x04C11DB7, MSB first, initial value 0, no final xor; the
                   muxer's byte-swapped state written little-endian = the conventional value written big-endian
  time base        choose_timebase :47-58 from the encoder's 1/25: 1/51200; one frame = 2048 ticks

Nothing here is product code; it pins the oracle (and through it the HIP path) to reference-held checksums."""
import hashlib

ID_STRING = b"nut/multimedia container\0"
MAIN, STREAM, SYNCPOINT, INDEX, INFO = (
    0x7A561F5F04AD + ((ord('N') << 8) + ord('M') << 48), 0x11405BF2F9DB + ((ord('N') << 8) + ord('S') << 48),
    0xE4ADEECA4569 + ((ord('N') << 8) + ord('K') << 48), 0xDD672F23E64E + ((ord('N') << 8) + ord('X') << 48),
    0xAB68B596BA78 + ((ord('N') << 8) + ord('I') << 48))
FLAG_KEY, FLAG_CODED_PTS, FLAG_STREAM_ID, FLAG_SIZE_MSB, FLAG_CHECKSUM, FLAG_HEADER_IDX, FLAG_CODED, FLAG_INVALID = 1, 8, 16, 32, 64, 1024, 4096, 8192
MAX_DISTANCE = 1024 * 32 - 1

_CRC = []
for _i in range(256):
    _c = _i << 24
    for _ in range(8):
        _c = ((_c << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if _c & 0x80000000 else (_c << 1) & 0xFFFFFFFF
    _CRC.append(_c)


def crc(data):
    c = 0
    for b in data:
        c = (_CRC[(c >> 24) ^ b] ^ (c << 8)) & 0xFFFFFFFF
    return c.to_bytes(4, "big")


def v(val):
    """put_v :321-329"""
    n = 1
    t = val
    while t >> 7:
        t >>= 7
        n += 1
    out = bytearray()
    for i in range(n - 1, 0, -1):
        out.append(128 | ((val >> (7 * i)) & 0x7F))
    out.append(val & 127)
    return bytes(out)


def s(val):
    """put_s :345-348"""
    return v(2 * abs(val) - (1 if val > 0 else 0))


def string(t):
    return v(len(t)) + t


def packet(startcode, payload):
    """put_packet :351-367"""
    forw = len(payload) + 4
    head = startcode.to_bytes(8, "big") + v(forw)
    if forw > 4096:
        head += crc(head)
    return head + payload + crc(payload)


class FrameCode:
    def __init__(self):
        self.flags = self.pts_delta = self.stream_id = self.size_mul = self.size_lsb = self.header_idx = 0


ELISION = [b"", b"\x00\x00\x01", b"\x00\x00\x01\xB6", b"\xFF\xFA", b"\xFF\xFB", b"\xFF\xFC", b"\xFF\xFD"]     # :150-163


def build_frame_code(frame_size):
    """:167-308 for nb_streams = 1, video, video_delay = 0"""
    fc = [FrameCode() for _ in range(257)]
    start, end = 1, 254
    fc[start].flags, fc[start].size_mul, fc[start].pts_delta = FLAG_CODED, 1, 1
    start += 1
    start2, end2 = start, end                          # the one stream owns [start, end)
    for key_frame in (0, 1):
        f = fc[start2]
        f.flags = FLAG_KEY * key_frame | FLAG_SIZE_MSB | FLAG_CODED_PTS
        f.stream_id, f.size_mul = 0, 1
        start2 += 1
    f = fc[start2]
    f.flags, f.stream_id, f.size_mul, f.pts_delta = FLAG_KEY | FLAG_SIZE_MSB, 0, 1, frame_size
    start2 += 1
    pred_table = [1 * frame_size]
    for pred in range(1):
        start3 = start2 + (end2 - start2) * pred // 1
        end3 = start2 + (end2 - start2) * (pred + 1) // 1
        for index in range(start3, end3):
            f = fc[index]
            f.flags = FLAG_SIZE_MSB                    # key_frame = intra_only = 0 for video
            f.stream_id, f.size_mul, f.size_lsb, f.pts_delta = 0, end3 - start3, index - start3, pred_table[pred]
    # memmove(&frame_code['N' + 1], &frame_code['N'], (255 - 'N') entries): shift up by one from 'N'
    n = ord('N')
    moved = fc[:n] + [FrameCode()] + fc[n:255]
    fc = moved[:256]
    for i in (0, 255, n):
        fc[i] = FrameCode()
        fc[i].flags = FLAG_INVALID
    return fc


def main_header(fc, tb_num, tb_den):
    """write_mainheader :369-452, version 3"""
    o = bytearray()
    o += v(3) + v(1) + v(MAX_DISTANCE) + v(1) + v(tb_num) + v(tb_den)
    tmp_pts, tmp_mul, tmp_stream, tmp_head_idx = 0, 1, 0, 0
    tmp_match = 1 - (1 << 62)
    i = 0
    while i < 256:
        tmp_fields, tmp_size = 0, 0
        if tmp_pts != fc[i].pts_delta: tmp_fields = 1
        if tmp_mul != fc[i].size_mul: tmp_fields = 2
        if tmp_stream != fc[i].stream_id: tmp_fields = 3
        if tmp_size != fc[i].size_lsb: tmp_fields = 4
        if tmp_head_idx != fc[i].header_idx: tmp_fields = 8
        tmp_pts, tmp_flags, tmp_stream = fc[i].pts_delta, fc[i].flags, fc[i].stream_id
        tmp_mul, tmp_size, tmp_head_idx = fc[i].size_mul, fc[i].size_lsb, fc[i].header_idx
        j = 0
        while i < 256:
            if i == ord('N'):
                i += 1                                  # j-- ; continue (with the loop's j++, i++)
                continue
            f = fc[i]
            if (f.pts_delta != tmp_pts or f.flags != tmp_flags or f.stream_id != tmp_stream or f.size_mul != tmp_mul or
                    f.size_lsb != tmp_size + j or f.header_idx != tmp_head_idx):
                break
            j += 1
            i += 1
        if j != tmp_mul - tmp_size:
            tmp_fields = 6
        o += v(tmp_flags) + v(tmp_fields)
        if tmp_fields > 0: o += s(tmp_pts)
        if tmp_fields > 1: o += v(tmp_mul)
        if tmp_fields > 2: o += v(tmp_stream)
        if tmp_fields > 3: o += v(tmp_size)
        if tmp_fields > 4: o += v(0)
        if tmp_fields > 5: o += v(j)
        if tmp_fields > 6: o += v(tmp_match & ((1 << 64) - 1))
        if tmp_fields > 7: o += v(tmp_head_idx)
    o += v(len(ELISION) - 1)
    for h in ELISION[1:]:
        o += v(len(h)) + h
    return bytes(o)


def stream_header(fourcc, width, height, msb_pts_shift, max_pts_distance):
    """write_streamheader :454-507: video, no extradata, unknown sample aspect ratio"""
    return (v(0) + v(0) + v(4) + fourcc + v(0) + v(msb_pts_shift) + v(max_pts_distance) + v(0) + b"\x00" + v(0) +
            v(width) + v(height) + v(0) + v(0) + v(0))


def mux(frames, width, height, fourcc, fps=25):
    """the bytes `ffmpeg ... -vcodec rawvideo -f nut` writes for these frames (bit-exact flags: no encoder tag)"""
    # choose_timebase(1/fps, 48000)
    num, den = 1, fps
    while den // num < 48000 and den < (1 << 24):
        den <<= 1
    frame_size = den // fps                              # av_div_q(1/fps, tb): integral here
    msb_pts_shift = 7 if 1000 * num >= den else 14
    max_pts_distance = max(den, num) // num
    fc = build_frame_code(frame_size)
    out = bytearray(ID_STRING)
    out += packet(MAIN, main_header(fc, num, den))
    out += packet(STREAM, stream_header(fourcc, width, height, msb_pts_shift, max_pts_distance))
    out += packet(INFO, v(0) + v(0) + v(0) + v(0) + v(0))                                   # write_globalinfo: no entries
    # write_streaminfo :546-583: the stream's metadata — ffmpeg's set_encoder_id (fftools/ffmpeg.c:2826-2865) writes
    # "Lavc <encoder>" without a version under the bit-exact flags — then r_frame_rate from avg_frame_rate
    out += packet(INFO, v(1) + v(0) + v(0) + v(0) + v(2) + string(b"encoder") + s(-1) + string(b"Lavc rawvideo") +
                  string(b"r_frame_rate") + s(-1) + string(b"%d/1" % fps))
    sp_pos = []                                          # syncpoint positions
    index_entries = []                                   # (pts, syncpoint pos) of key frames
    keyframe_pts = {}
    last_flags = 0
    for k, data in enumerate(frames):
        pts = dts = k * frame_size
        # nut_write_packet: every raw frame is a key frame and larger than max_distance -> store_sp
        back = 0
        cands = [p for (t, p) in index_entries if t <= dts]
        pos = len(out)
        if cands:
            back = (pos - cands[-1]) >> 4
        out += packet(SYNCPOINT, v(dts * 1 + 0) + v(back))
        sp_pos.append(pos)
        last_pts = dts                                   # ff_nut_reset_ts
        coded_pts = pts & ((1 << msb_pts_shift) - 1)
        mask = (1 << msb_pts_shift) - 1
        delta = last_pts - mask // 2
        if ((coded_pts - delta) & mask) + delta != pts:
            coded_pts = pts + (1 << msb_pts_shift)
        size = len(data)

        def needed(f):
            fl = FLAG_KEY
            if 0 != f.stream_id: fl |= FLAG_STREAM_ID
            if size // f.size_mul: fl |= FLAG_SIZE_MSB
            if pts - last_pts != f.pts_delta: fl |= FLAG_CODED_PTS
            if size > 2 * MAX_DISTANCE: fl |= FLAG_CHECKSUM
            if abs(pts - last_pts) > max_pts_distance: fl |= FLAG_CHECKSUM
            return fl | (f.flags & FLAG_CODED)
        best_len, code = 1 << 30, -1
        for i in range(256):
            f = fc[i]
            flags = f.flags
            if flags & FLAG_INVALID:
                continue
            nf = needed(f)
            length = 0
            if flags & FLAG_CODED:
                length += 1
                flags = nf
            if (flags & nf) != nf: continue
            if (flags ^ nf) & FLAG_KEY: continue
            if flags & FLAG_STREAM_ID: length += len(v(0))
            if size % f.size_mul != f.size_lsb: continue
            if flags & FLAG_SIZE_MSB: length += len(v(size // f.size_mul))
            if flags & FLAG_CHECKSUM: length += 4
            if flags & FLAG_CODED_PTS: length += len(v(coded_pts))
            # header_len[best_header_idx = 0] = 0: no elision
            length -= 0
            length *= 4
            length += 0 if flags & FLAG_CODED_PTS else 1
            length += 0 if flags & FLAG_CHECKSUM else 1
            if length < best_len:
                best_len, code = length, i
        f = fc[code]
        flags, nf = f.flags, needed(f)
        head = bytearray([code])
        if flags & FLAG_CODED:
            head += v((flags ^ nf) & ~FLAG_CODED)
            flags = nf
        if flags & FLAG_STREAM_ID: head += v(0)
        if flags & FLAG_CODED_PTS: head += v(coded_pts)
        if flags & FLAG_SIZE_MSB: head += v(size // f.size_mul)
        if flags & FLAG_CHECKSUM: head += crc(head)
        out += head + data
        last_flags = flags
        index_entries.append((pts, pos))
        keyframe_pts.setdefault(len(sp_pos), pts)        # keyframe_pts[sp_count] = pts
    # ---- write_index :605-662 -------------------------------------------------------------------------------------
    sp_count = len(sp_pos)
    if sp_count:
        idx = bytearray()
        max_pts = (len(frames) - 1) * frame_size
        idx += v(max_pts * 1 + 0) + v(sp_count)
        prev = 0
        for p in sp_pos:
            idx += v((p >> 4) - (prev >> 4))
            prev = p
        NOPTS = None
        kp = [keyframe_pts.get(j, NOPTS) for j in range(2 * sp_count + 2)]
        last = -1
        j = 0
        while j < sp_count:
            n = 0
            flag = (kp[j] is not NOPTS) ^ (j + 1 == sp_count)
            while j < sp_count and (kp[j] is not NOPTS) == flag:
                n += 1
                j += 1
            idx += v(1 + 2 * int(flag) + 4 * n)
            for k in range(j - n, min(j, sp_count - 1) + 1):
                if k >= sp_count or kp[k] is NOPTS:
                    continue
                idx += v(kp[k] - last)
                last = kp[k]
            j += 1                                      # the outer for's own j++ (the entry after a run is skipped)
        payload_size = len(idx) + 8 + 4
        import math
        log2 = payload_size.bit_length() - 1
        idx += (8 + payload_size + log2 // 7 + 1 + 4 * (payload_size > 4096)).to_bytes(8, "big")
        out += packet(INDEX, bytes(idx))
    return bytes(out)


def md5(frames, width, height, fourcc, fps=25):
    return hashlib.md5(mux(frames, width, height, fourcc, fps)).hexdigest()


FOURCC = {"yuv420p": b"I420", "nv12": b"NV12", "rgb24": b"RGB\x18", "bgr24": b"BGR\x18", "rgba": b"RGBA", "bgra": b"BGRA",
          "yuv444p": b"444P"}
