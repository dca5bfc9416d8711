/* oracle/drc_oracle.h — TEST INFRASTRUCTURE (see oracle_common.h).
 * CPU oracle for the Draco 2.2 triangular-mesh bitstream as emitted by
 * `draco_encoder -qp 11 -qt 10 -qn 8 -cl 7` (scripts/Encoder.py:260) and pinned by the
 * reference fixtures example/public/liam/output/geometry_draco/NNNNN.drc  (SURVEY Appendix A/C).
 */
#ifndef UVOL_DRC_ORACLE_H
#define UVOL_DRC_ORACLE_H
#include "oracle_common.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- entropy coders (SURVEY A.2 decode, A.10 / D.7 encode) ---------- */
/* Decode a DecodeSymbols() section starting at b[*o]; returns 0 ok. out must hold nvals. */
typedef struct { int scheme, bl, prec_bits, alphabet, unique, left; uint32_t final_state, base; size_t payload; } orc_sym_info;
int orc_decode_symbols(const uint8_t *b, size_t n, size_t *o, uint32_t nvals, uint32_t *out, orc_sym_info *info);
/* the same for nvals values in groups of ncomp (the TAGGED scheme codes one bit-length tag per group); RAW ignores ncomp */
int orc_decode_symbols_nc(const uint8_t *b, size_t n, size_t *o, uint32_t nvals, int ncomp, uint32_t *out, orc_sym_info *info);
/* Encode symbols with the RAW rANS scheme exactly as draco's EncodeSymbols(RAW). */
void orc_encode_symbols(const uint32_t *syms, uint32_t nvals, orc_buf *out);

typedef struct { const uint8_t *buf; size_t off; uint32_t st; uint8_t p0; size_t end; } orc_rabs_dec;
int orc_rabs_open(orc_rabs_dec *r, const uint8_t *b, size_t n, size_t o);
int orc_rabs_bit(orc_rabs_dec *r);
/* bits: one byte per bit (0/1), encoded in order; emits p0, varint(len), payload */
void orc_rabs_encode(const uint8_t *bits, size_t nbits, orc_buf *out);

/* ---------- decoded mesh ---------- */
typedef struct {
  int att_type;      /* 0 POSITION, 1 NORMAL, 3 TEX_COORD, 4 GENERIC */
  int data_type, ncomp, unique_id;
  int dec_type;      /* 0 vertex attribute, 1 corner attribute */
  int att_data_id;   /* -1 for position */
  int seq_type;      /* 1 integer, 2 quantization, 3 normals */
  int pred_method, transform;
  int n;             /* number of entries */
  int ncomp_port;    /* components of the portable (integer) values: ncomp, or 2 for normals */
  int32_t *vals;     /* n * ncomp_port portable integer values, entry order */
  int32_t *corner_to_entry; /* 3*nf */
  float minv[4], range; int qbits;
  size_t sec_begin, sec_end;      /* byte span of values+prediction data */
  size_t sym_begin, sym_end;      /* byte span of the rANS symbol section */
  int n_orient, n_flip_set;
  uint32_t n_seam_corners;
} drc_att;

typedef struct {
  int major, minor, nf, nev, nad, nsym, nsplit, nts, nverts_alloc;
  int32_t *opp, *c2v;            /* 3*nf (NULL for sequential connectivity) */
  int ctx_n[6];
  int n_interior_start;
  size_t conn_end, hdr_end, total;
  int natt;
  drc_att att[8];
  size_t leftover;
  int method, traversal;         /* encoder_method 1 edgebreaker / 0 sequential; edgebreaker traversal 2 valence / 0 standard; sequential: connectivity method 0 compressed / 1 raw */
  int npoints;                   /* sequential: number of points (= entries of every attribute) */
} drc_mesh;

/* returns 0 on success, negative error code otherwise */
int drc_decode(const uint8_t *b, size_t n, drc_mesh *m);
void drc_mesh_free(drc_mesh *m);
/* dequantise attribute a into out (n*ncomp floats: pos/uv  min+q*range/maxq ; normals unit vectors (3 comps)) */
void drc_dequant(const drc_mesh *m, int a, float *out);

/* ---------- encoder restatement ---------- */
typedef struct {
  int qp, qt, qn;     /* quantization bits, defaults 11/10/8 (scripts/Encoder.py:171-173) */
  int method;         /* tool set: 0 = valence edgebreaker (draco_encoder -cl 7, the reference's default), 1 = edgebreaker with the STANDARD traversal
                         (symbols bit-coded, what stock levels 1..5 write), 2 = SEQUENTIAL connectivity + difference prediction (stock level 0) */
} drc_enc_params;

/* OBJ-style input: separate value arrays + per-corner indices. uv/nrm may be NULL (with n=0). */
typedef struct {
  const float *pos; uint32_t n_pos;
  const float *uv;  uint32_t n_uv;
  const float *nrm; uint32_t n_nrm;
  const uint32_t *idx_pos, *idx_uv, *idx_nrm;  /* 3*nf each */
  uint32_t nf;
} drc_enc_input;

int drc_encode(const drc_enc_input *in, const drc_enc_params *p, orc_buf *out);

#ifdef __cplusplus
}
#endif
#endif
