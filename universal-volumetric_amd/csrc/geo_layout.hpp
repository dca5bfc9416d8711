// geo_layout.hpp - bitstream layout and gather.
// Part of the geometry encoder translation unit: included by geom_encode.hip, in pipeline order (not a standalone header).
// ------------------------------------------------------------------------------------------------
// layout: small header pieces + piece list (single lane per frame), then a parallel gather
// ------------------------------------------------------------------------------------------------
__device__ inline void add_piece(GeoJob &J, const uint8_t *p, uint32_t len, uint32_t &total) {
  if (J.n_pieces >= GEO_MAXPIECES) { J.status = -40; return; }
  J.piece_ptr[J.n_pieces] = p; J.piece_len[J.n_pieces] = len; J.piece_off[J.n_pieces] = total; J.n_pieces++; total += len;
}
__device__ inline void put_i32(uint8_t *a, uint32_t &o, int32_t v) { for (int k = 0; k < 4; k++) a[o++] = (uint8_t)((uint32_t)v >> (8 * k)); }
__device__ inline void put_f32(uint8_t *a, uint32_t &o, float f) { uint32_t u; memcpy(&u, &f, 4); for (int k = 0; k < 4; k++) a[o++] = (uint8_t)(u >> (8 * k)); }
__device__ inline void add_rans(GeoJob &J, int s, uint32_t &total) {
  RansStream &S = J.rs[s];
  add_piece(J, S.head, S.head_len, total); add_piece(J, S.pay + S.pay_off, S.pay_len, total);
}
__global__ void __launch_bounds__(64) k_layout(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.x];
  if (threadIdx.x != 0 || J.status != 0) return;
  uint8_t *a = J.arena; uint32_t o = 0, total = 0, b0;
  J.n_pieces = 0;
  // header + connectivity header
  b0 = o;
  a[o++] = 'D'; a[o++] = 'R'; a[o++] = 'A'; a[o++] = 'C'; a[o++] = 'O'; a[o++] = 2; a[o++] = 2; a[o++] = 1; a[o++] = 1; a[o++] = 0; a[o++] = 0;
  a[o++] = 2;
  o += g_put_varint(a + o, J.nverts); o += g_put_varint(a + o, J.nf); a[o++] = (uint8_t)J.nad;
  o += g_put_varint(a + o, (uint32_t)J.nsym); o += g_put_varint(a + o, (uint32_t)J.nsplit);
  o += g_put_varint(a + o, (uint32_t)J.nev);
  { int last = 0;
    if (o + 10 * (uint32_t)J.nev + 64 > J.arena_cap) { J.status = -41; return; }
    for (int i = 0; i < J.nev; i++) { o += g_put_varint(a + o, (uint32_t)(J.ev_src[i] - last)); o += g_put_varint(a + o, (uint32_t)(J.ev_src[i] - J.ev_spl[i])); last = J.ev_src[i]; }
    if (J.nev > 0) { int nb = (J.nev + 7) / 8; for (int j = 0; j < nb; j++) { uint8_t v = 0; for (int k = 0; k < 8 && 8 * j + k < J.nev; k++) v |= (uint8_t)((J.ev_edge[8 * j + k] & 1) << k); a[o++] = v; } } }
  add_piece(J, a + b0, o - b0, total);
  add_piece(J, J.rb[0].buf + J.rb[0].off, J.rb[0].len, total);
  for (int i = 0; i < J.nad; i++) add_piece(J, J.rb[1 + i].buf + J.rb[1 + i].off, J.rb[1 + i].len, total);
  for (int i = 0; i < 6; i++) {
    b0 = o; o += g_put_varint(a + o, J.ctx_n[i]); add_piece(J, a + b0, o - b0, total);
    if (J.ctx_n[i] > 0) add_rans(J, i, total);
  }
  // attribute decoder headers
  b0 = o;
  const int dec_type[2] = { J.interior_seams[0] ? 1 : 0, J.interior_seams[1] ? 1 : 0 };
  a[o++] = (uint8_t)(1 + J.nad);
  a[o++] = 0xff; a[o++] = 0; a[o++] = 0;
  for (int i = 0; i < J.nad; i++) { a[o++] = (uint8_t)i; a[o++] = (uint8_t)dec_type[i]; a[o++] = 0; }
  a[o++] = 1; a[o++] = 0; a[o++] = 9; a[o++] = 3; a[o++] = 0; a[o++] = 0; a[o++] = 2;
  for (int i = 0; i < J.nad; i++) {
    a[o++] = 1;
    if (J.att_kind[i] == 0) { a[o++] = 3; a[o++] = 9; a[o++] = 2; a[o++] = 0; a[o++] = (uint8_t)(1 + i); a[o++] = 2; }
    else { a[o++] = 1; a[o++] = 9; a[o++] = 3; a[o++] = 0; a[o++] = (uint8_t)(1 + i); a[o++] = 3; }
  }
  // position values
  a[o++] = 1; a[o++] = 1; a[o++] = 1;
  add_piece(J, a + b0, o - b0, total);
  add_rans(J, 6, total);
  b0 = o;
  put_i32(a, o, J.wrap_lo[0]); put_i32(a, o, J.wrap_hi[0]);
  for (int k = 0; k < 3; k++) put_f32(a, o, g_float_unorder(J.pos_min_u[k]));
  put_f32(a, o, quant_range(J.pos_min_u, J.pos_max_u, 3)); a[o++] = (uint8_t)J.qp;
  for (int i = 0; i < J.nad; i++) {
    if (J.att_kind[i] == 0) {
      a[o++] = 5; a[o++] = 1; a[o++] = 1;
      add_piece(J, a + b0, o - b0, total);
      add_rans(J, 7, total);
      b0 = o; put_i32(a, o, (int32_t)J.n_ori); add_piece(J, a + b0, o - b0, total);
      add_piece(J, J.rb[3].buf + J.rb[3].off, J.rb[3].len, total);
      b0 = o;
      put_i32(a, o, J.wrap_lo[1]); put_i32(a, o, J.wrap_hi[1]);
      put_f32(a, o, g_float_unorder(J.uv_min_u[0])); put_f32(a, o, g_float_unorder(J.uv_min_u[1]));
      put_f32(a, o, quant_range(J.uv_min_u, J.uv_max_u, 2)); a[o++] = (uint8_t)J.qt;
    } else {
      const GOct ot = g_oct(J.qn);
      a[o++] = 6; a[o++] = 3; a[o++] = 1;
      add_piece(J, a + b0, o - b0, total);
      add_rans(J, 8, total);
      b0 = o; put_i32(a, o, ot.MAXQ); put_i32(a, o, ot.CEN); add_piece(J, a + b0, o - b0, total);
      add_piece(J, J.rb[4].buf + J.rb[4].off, J.rb[4].len, total);
      b0 = o; a[o++] = (uint8_t)J.qn;
    }
  }
  add_piece(J, a + b0, o - b0, total);
  J.out_len = total;
  if (total > J.out_cap) J.status = UVOL_E_NOSPACE;
}
// layout of a frame with sequential connectivity (see k_sq_*): header, index section, ONE attributes decoder
__global__ void __launch_bounds__(64) k_sq_layout(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.x];
  if (threadIdx.x != 0 || J.status != 0) return;
  uint8_t *a = J.arena; uint32_t o = 0, total = 0, b0 = 0;
  J.n_pieces = 0;
  a[o++] = 'D'; a[o++] = 'R'; a[o++] = 'A'; a[o++] = 'C'; a[o++] = 'O'; a[o++] = 2; a[o++] = 2; a[o++] = 1; a[o++] = 0; a[o++] = 0; a[o++] = 0;
  o += g_put_varint(a + o, J.nf_in); o += g_put_varint(a + o, J.sq_np); a[o++] = 1;                 // connectivity_method 1: indices stored directly
  add_piece(J, a + b0, o - b0, total);
  add_piece(J, J.sq_idx, J.sq_idx_bytes, total);
  b0 = o;
  a[o++] = 1;
  o += g_put_varint(a + o, (uint32_t)(1 + J.nad));
  a[o++] = 0; a[o++] = 9; a[o++] = 3; a[o++] = 0; a[o++] = 0;
  { int id = 1;
    if (J.has_uv) { a[o++] = 3; a[o++] = 9; a[o++] = 2; a[o++] = 0; a[o++] = (uint8_t)id++; }
    if (J.has_nrm) { a[o++] = 1; a[o++] = 9; a[o++] = 3; a[o++] = 0; a[o++] = (uint8_t)id++; } }
  a[o++] = 2; if (J.has_uv) a[o++] = 2; if (J.has_nrm) a[o++] = 3;
  a[o++] = 0; a[o++] = 1; a[o++] = 1;                                                                // position: DIFFERENCE, wrap, compressed
  add_piece(J, a + b0, o - b0, total);
  add_rans(J, 6, total);
  b0 = o; put_i32(a, o, J.wrap_lo[0]); put_i32(a, o, J.wrap_hi[0]);
  if (J.has_uv) {
    a[o++] = 0; a[o++] = 1; a[o++] = 1;
    add_piece(J, a + b0, o - b0, total);
    add_rans(J, 7, total);
    b0 = o; put_i32(a, o, J.wrap_lo[1]); put_i32(a, o, J.wrap_hi[1]);
  }
  if (J.has_nrm) {
    const GOct ot = g_oct(J.qn);
    a[o++] = 0; a[o++] = 3; a[o++] = 1;
    add_piece(J, a + b0, o - b0, total);
    add_rans(J, 8, total);
    b0 = o; put_i32(a, o, ot.MAXQ); put_i32(a, o, ot.CEN);
  }
  for (int k = 0; k < 3; k++) put_f32(a, o, g_float_unorder(J.pos_min_u[k]));
  put_f32(a, o, quant_range(J.pos_min_u, J.pos_max_u, 3)); a[o++] = (uint8_t)J.qp;
  if (J.has_uv) { put_f32(a, o, g_float_unorder(J.uv_min_u[0])); put_f32(a, o, g_float_unorder(J.uv_min_u[1])); put_f32(a, o, quant_range(J.uv_min_u, J.uv_max_u, 2)); a[o++] = (uint8_t)J.qt; }
  if (J.has_nrm) a[o++] = (uint8_t)J.qn;
  add_piece(J, a + b0, o - b0, total);
  J.out_len = total;
  if (total > J.out_cap) J.status = UVOL_E_NOSPACE;
}
// the frames' bitstreams are gathered back to back (16-byte aligned) so that the host fetches the whole batch with ONE copy
__global__ void __launch_bounds__(64) k_out_offsets(GeoJob *jobs, int n) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint64_t off = 0;
  for (int i = 0; i < n; i++) {
    jobs[i].out_pack_off = off;
    if (jobs[i].status != 0) continue;
    const uint64_t len = ((uint64_t)jobs[i].out_len + 15) & ~(uint64_t)15;
    if (off + len > jobs[i].slab_cap) { jobs[i].status = GEO_E_SLAB_FULL; continue; }      // the packed area is sized for typical streams
    off += len;
  }
}
__global__ void __launch_bounds__(UVOL_BLOCK) k_gather(GeoJob *jobs) {
  GeoJob &J = jobs[blockIdx.z];
  if (J.status != 0) return;
  const uint32_t pc = blockIdx.y;
  if (pc >= J.n_pieces) return;
  const uint8_t *src = J.piece_ptr[pc]; uint8_t *dst = J.out_pack + J.out_pack_off + J.piece_off[pc]; const uint32_t len = J.piece_len[pc];
  for (uint32_t i = blockIdx.x * UVOL_BLOCK + threadIdx.x; i < len; i += gridDim.x * UVOL_BLOCK) dst[i] = src[i];
}

