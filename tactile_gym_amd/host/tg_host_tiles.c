/* tg_host_tiles.c - host side of the tile-sparse observation download (plain C, no device code).
 *
 * The numpy VecEnv boundary (what the reference's sb3_helpers consume: stable-baselines3 VecEnv.step_wait() -> numpy observations,
 * sb3_helpers/rl_utils.py:17-30) copies the whole uint8 batch device -> host every step (16.8 MB for 1024 x 128 x 128).  With the tile
 * payload of csrc/tg_exchange.hip (tg_pack_tiles: only the 16 x 16 tiles that differ from the untouched sensor's image, 272-byte records
 * behind a 16-byte header) 1/10 of that crosses PCIe, and this function rebuilds the batch in a PERSISTENT host buffer the way
 * tg_unpack_tiles_multi does on rank 0's device: the tiles this buffer's previous frame had live get the template back, then the new
 * records land.  Message layout (parallel.py: TILE_MAGIC, TILE_REC): int32 {count, n_images, tiles_per_image, 0x54475431}, then count records
 * of {int32 tile id = image * tiles_per_image + tile, 12 bytes unused, 256 pixel bytes (16 rows x 16)}.
 */
#include <stdint.h>
#include <string.h>

#define TG_TILE_MAGIC 0x54475431
#define TG_TILE_REC 272

/* msg: the message (msg_bytes available); tmpl: uint8 [H*W]; dst: uint8 [n][H][W], holding what the previous call on it left;
 * prev_ids: int32 [n * (H/16) * (W/16)] capacity, *n_prev entries valid on entry (the tiles of dst that differ from tmpl), the new list on
 * return.  Returns the record count, or -1 bad argument, -2 bad header, -3 message shorter than its count says, -4 tile id out of range. */
int64_t tg_host_unpack_tiles(const uint8_t* msg, int64_t msg_bytes, const uint8_t* tmpl, int32_t n, int32_t H, int32_t W, uint8_t* dst,
                             int32_t* prev_ids, int64_t* n_prev) {
    if (!msg || !tmpl || !dst || !prev_ids || !n_prev || n <= 0 || H <= 0 || W <= 0 || (H & 15) || (W & 15) || msg_bytes < 16) return -1;
    const int32_t TW = W / 16, T = (H / 16) * TW;
    int32_t hdr[4];
    memcpy(hdr, msg, 16);
    const int64_t count = hdr[0], total = (int64_t)n * T;
    if ((uint32_t)hdr[3] != (uint32_t)TG_TILE_MAGIC || hdr[1] != n || hdr[2] != T || count < 0 || count > total) return -2;
    if (16 + count * TG_TILE_REC > msg_bytes) return -3;
    if (*n_prev < 0 || *n_prev > total) return -1;
    const size_t img_bytes = (size_t)H * W;
    for (int64_t k = 0; k < *n_prev; ++k) {          /* the previous frame's live tiles: back to the untouched sensor's image */
        const int32_t id = prev_ids[k];
        if (id < 0 || id >= total) return -4;
        const int32_t img = id / T, tile = id % T, ty = tile / TW, tx = tile % TW;
        const uint8_t* s = tmpl + (size_t)ty * 16 * W + (size_t)tx * 16;
        uint8_t* d = dst + (size_t)img * img_bytes + (size_t)ty * 16 * W + (size_t)tx * 16;
        for (int r = 0; r < 16; ++r) memcpy(d + (size_t)r * W, s + (size_t)r * W, 16);
    }
    const uint8_t* rec = msg + 16;
    for (int64_t k = 0; k < count; ++k, rec += TG_TILE_REC) {
        int32_t id;
        memcpy(&id, rec, 4);
        if (id < 0 || id >= total) return -4;
        const int32_t img = id / T, tile = id % T, ty = tile / TW, tx = tile % TW;
        uint8_t* d = dst + (size_t)img * img_bytes + (size_t)ty * 16 * W + (size_t)tx * 16;
        for (int r = 0; r < 16; ++r) memcpy(d + (size_t)r * W, rec + 16 + 16 * r, 16);
        prev_ids[k] = id;
    }
    *n_prev = count;
    return count;
}

/* uint8 [n][H][W] <- n copies of tmpl: the state a fresh buffer (no live tiles) must have. */
int32_t tg_host_fill_template(const uint8_t* tmpl, int32_t n, int32_t H, int32_t W, uint8_t* dst) {
    if (!tmpl || !dst || n <= 0 || H <= 0 || W <= 0) return -1;
    const size_t img_bytes = (size_t)H * W;
    for (int32_t i = 0; i < n; ++i) memcpy(dst + (size_t)i * img_bytes, tmpl, img_bytes);
    return 0;
}
