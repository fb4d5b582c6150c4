

# ------------------------------------------------------------------------------------------------------------
# 16-sample-tile engine (csrc/mlp_engine16.h): output tiles of 16 rows, k-steps of 32, k-slot kappa = 32ks + 8q + j
# holds feature 32ks + 16(j>>2) + 4q + (j&3)
# ------------------------------------------------------------------------------------------------------------
def slot_features16(ks_count):
    out = np.empty(ks_count * 32, dtype=np.int64)
    for ks in range(ks_count):
        for q in range(4):
            for j in range(8):
                out[32 * ks + 8 * q + j] = 32 * ks + 16 * (j >> 2) + 4 * q + (j & 3)
    return out


def build_sdf_jobs16():
    ly = query_layout(2)
    offW, offb, total = flat_offsets(SDF_IN, SDF_OUT)
    offB, offG, offV, total_raw = raw_offsets(SDF_IN, SDF_OUT)
    row_tab, rbase = row_table(SDF_IN, SDF_OUT, offW)
    b = _Builder(geom=1)
    inv_sqrt2 = 1.0 / math.sqrt(2.0)
    for l in range(9):
        ksf, ntf, ksr, ntr = [int(v) for v in ly.geom[l]]
        fwd_hi, fwd_lo, rev_hi, rev_lo, bias = [int(v) for v in ly.off[l]]
        n_in, n_out = SDF_IN[l], SDF_OUT[l]
        scale = inv_sqrt2 if l == 4 else 1.0
        rws = np.arange(ntf * 16)
        if l == 8:
            rowmap = np.where(rws < 256, rws + 1, np.where(rws == 256, 0, -1))
        else:
            rowmap = _lim(rws, n_out)
        feat = slot_features16(ksf)
        if l == 4:   # 7 k-steps of h (224 slots, 217 valid) then 2 k-steps of PE (64 slots, 39 valid)
            kmap = np.where(feat < 224, _lim(feat, 217), np.where(feat - 224 < N_PE, 217 + (feat - 224), -1))
        else:
            kmap = _lim(feat, n_in)
        b.frag(fwd_hi, fwd_lo, offV[l], n_in, ksf, ntf, 0, rowmap, kmap, scale, rbase[l])
        ofeat = slot_features16(ksr)
        if l == 8:
            kmap_r = np.where(ofeat < 256, ofeat + 1, np.where(ofeat == 256, 0, -1))
        else:
            kmap_r = _lim(ofeat, n_out)
        rin = np.arange(ntr * 16)
        if l == 4:   # 14 row tiles of h, then 3 row tiles of PE
            rowmap_r = np.where(rin < 224, _lim(rin, 217), np.where(rin - 224 < N_PE, 217 + (rin - 224), -1))
        else:
            rowmap_r = _lim(rin, n_in)
        b.frag(rev_hi, rev_lo, offV[l], n_in, ksr, ntr, 1, rowmap_r, kmap_r, scale, rbase[l])
        b.accvec(bias, offB[l], 1, ntf, rowmap)
    b.accvec(ly.extra, offV[8], 1, 16, np.arange(256), rs_base=rbase[8], rs_mode=2)
    jobs, maps, units = b.finish()
    segs = np.array([(offb[l], offB[l], SDF_OUT[l], 0) for l in range(9)], dtype=np.int32)
    return {"layout": ly, "jobs": jobs, "maps": maps, "units": units, "n_params": total, "offW": offW, "offb": offb,
            "ins": SDF_IN, "outs": SDF_OUT, "n_raw": total_raw, "offB": offB, "offG": offG, "offV": offV,
            "rows": row_tab, "bias_segs": segs}


def build_color_jobs16():
    ly = query_layout(3)
    offW, offb, total = flat_offsets(COL_IN, COL_OUT)
    offB, offG, offV, total_raw = raw_offsets(COL_IN, COL_OUT)
    row_tab, rbase = row_table(COL_IN, COL_OUT, offW)
    b = _Builder(geom=1)
    for l in range(5):
        ksf, ntf, ksr, ntr = [int(v) for v in ly.geom[l]]
        fwd_hi, fwd_lo, rev_hi, rev_lo, bias = [int(v) for v in ly.off[l]]
        n_in, n_out = COL_IN[l], COL_OUT[l]
        rowmap = _lim(np.arange(ntf * 16), n_out)
        feat = slot_features16(ksf)
        if l == 0:
            kmap = np.where(feat < 256, feat + N_SIDE, np.where(feat - 256 < N_SIDE, feat - 256, -1))
        else:
            kmap = _lim(feat, n_in)
        b.frag(fwd_hi, fwd_lo, offV[l], n_in, ksf, ntf, 0, rowmap, kmap, 1.0, rbase[l])
        kmap_r = _lim(slot_features16(ksr), n_out)
        rin = np.arange(ntr * 16)
        if l == 0:
            rowmap_r = np.where(rin < 256, rin + N_SIDE, np.where(rin - 256 < N_SIDE, rin - 256, -1))
        else:
            rowmap_r = _lim(rin, n_in)
        b.frag(rev_hi, rev_lo, offV[l], n_in, ksr, ntr, 1, rowmap_r, kmap_r, 1.0, rbase[l])
        b.accvec(bias, offB[l], 1, ntf, rowmap)
    jobs, maps, units = b.finish()
    segs = np.array([(offb[l], offB[l], COL_OUT[l], 0) for l in range(5)], dtype=np.int32)
    return {"layout": ly, "jobs": jobs, "maps": maps, "units": units, "n_params": total, "offW": offW, "offb": offb,
            "ins": COL_IN, "outs": COL_OUT, "n_raw": total_raw, "offB": offB, "offG": offG, "offV": offV,
            "rows": row_tab, "bias_segs": segs}
