// spdp_records.cpp -- the exon-form result records of an alignment (host only, no device work).
//
// What Gsinfo::ExonForm (src/sqpr.cc:820-996) derives from the per-exon EISCR records skl_rngS_ng / skl_rngH_ng leave
// behind (here: SpdpRescored::exons, from spdp_skl_rng_s / _h): one ExonRecord per exon and one GeneRecord per
// alignment -- the payload of the -O12 `.erd` / `.grd` files sortgrcd reads (struct layouts of src/seq.h:1212-1255) --
// and the same numbers as the text lines of -O4.  Frame-shift records (iscr = NEVSEL) split an exon without ending it.
#include "../../include/spdp.h"
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

const int NEV = SPDP_NEVSEL;
const char NUCL[] = "--ACMGRSVTWYHKDBN";                        // src/seq.cc:56
const char NCODON[] = "--NCGAAGAAGATTATTCCCGATGGA";              // src/seq.cc:60

inline int site(const SpdpSiteMap& m, int n) { return m.site0 + n * m.step; }         // Seq::SiteNo
inline bool live(const SpdpExon& e) { return e.left != INT32_MAX; }                    // neoeij

struct Walk {
    std::vector<SpdpExonRecord> ex;
    std::vector<int> mmc, unp;          // per exon, for the text form
    SpdpGeneRecord g;
};

// one pass over the records: an exon closes at every record that carries an intron score (the last one carries 0);
// what describes the intron in front of an exon (length, phase, dinucleotides, frame-shift slack) is filled in when the
// previous exon closes and rides on the next record
int walk(const SpdpExonFormIn& in, Walk& w)
{
    if (!in.eij || in.n_eij < 1 || !in.gene_codes) return -1;
    const SpdpExon* rec = in.eij;
    const SpdpExon* const end = rec + in.n_eij;
    const int per_res = in.qry_is_protein ? 3 : 1;
    const char* letters = in.gene_is_tron ? NCODON : NUCL;
    const size_t n_letters = in.gene_is_tron ? sizeof NCODON - 1 : sizeof NUCL - 1;
    auto letter = [&](int pos) { const unsigned c = in.gene_codes[pos]; return c < n_letters ? letters[c] : '?'; };
    memset(&w.g, 0, sizeof w.g);
    SpdpExonRecord er;
    memset(&er, 0, sizeof er);
    int covered = 0, matched = 0, prev_iscr = 0;
    const SpdpExon* first = rec;
    const SpdpExon* prev = rec;          // the exon closed last
    const SpdpExon* open = rec;          // where the current exon began
    while (rec < end && live(*rec)) {
        if (rec->iscr > NEV) {
            if (w.g.nexn) { w.g.bmmc += prev->mmc3 + rec->mmc5; w.g.bunp += prev->unp3 + rec->unp5; }
            const int span = rec->right - open->left;
            const int rlen = (rec->rright - open->rleft) * in.q_many + rec->unp;
            covered += span; matched += rec->mch;
            w.g.mmc += rec->mmc; w.g.unp += rec->unp; ++w.g.nexn;
            er.Pmatch = rlen ? (float) (100. * rec->mch / rlen) : 0.f;
            er.Elen = span;
            er.Nmmc = rec->mmc; er.Nunp = rec->unp;
            er.Rleft = site(in.qmap, open->rleft); er.Rright = site(in.qmap, rec->rright - 1);
            er.Gleft = site(in.gmap, open->left); er.Gright = site(in.gmap, rec->right - 1);
            er.Bmmc = prev->mmc3 + rec->mmc5;
            er.Bunp = (prev->unp3 % per_res || rec->unp5 % per_res) ? 9 : (prev->unp3 + rec->unp5) / per_res;
            er.Escore = (float) rec->escr / in.scale;
            er.Iscore = (float) prev_iscr / in.scale;
            er.Sig3 = (float) rec->sig3 / in.scale;
            er.Sig5 = (float) rec->sig5 / in.scale;
            w.ex.push_back(er); w.mmc.push_back(rec->mmc); w.unp.push_back(rec->unp);
            prev_iscr = rec->iscr;
            prev = rec++;
            if (!(rec < end && live(*rec))) break;
            open = rec;
            er.Ilen = rec->left - prev->right;
            er.phase = in.qry_is_protein ? (3 - prev->phs) % 3 : covered % 3;
            er.miss = 0;
            er.Iends[0] = letter(prev->right); er.Iends[1] = letter(prev->right + 1);
            er.Iends[2] = letter(rec->left - 2); er.Iends[3] = letter(rec->left - 1);
        } else {                        // a frame shift inside the exon
            ++w.g.ng;
            ++rec;
            if (!(rec < end && live(*rec))) break;
            er.miss = rec->left - open->right;
        }
    }
    const int qspan = in.q_right - in.q_left;
    w.g.Gstart = site(in.gmap, first->left);
    w.g.Gend = site(in.gmap, prev->right - 1);
    w.g.Rstart = site(in.qmap, in.q_left);
    w.g.Rend = site(in.qmap, in.q_right - 1);
    w.g.Gscore = in.scr / in.aln_scale;
    w.g.Pmatch = (float) (100. * matched / qspan);
    w.g.Pcover = (float) (100. * (matched + w.g.mmc) / qspan);
    w.g.Csense = w.g.Gstart > w.g.Gend;
    w.g.Cid = in.gene_id;
    w.g.Rid = in.qry_id;
    w.g.Rsense = (short) in.q_sens;
    w.g.Rlen = in.q_len;
    w.g.Nrecord = (uint32_t) in.first_exon_record;
    return 0;
}

}   // namespace

extern "C" int spdp_exon_form(const SpdpExonFormIn* in, SpdpExonRecord* exons, int cap, SpdpGeneRecord* gene)
{
    if (!in || !gene) return -1;
    Walk w;
    if (walk(*in, w)) return -1;
    if ((int) w.ex.size() > cap || (!exons && !w.ex.empty())) return -1;
    for (size_t i = 0; i < w.ex.size(); ++i) exons[i] = w.ex[i];
    *gene = w.g;
    if (in->qry_is_protein && gene->ng == 0) gene->ng = -1;       // (the binary form only: sqpr.cc:966)
    return (int) w.ex.size();
}

extern "C" int spdp_exon_form_text(const SpdpExonFormIn* in, const char* qname, const char* gname, int header,
                                   char* buf, int cap)
{
    if (!in || !qname || !gname || !buf || cap < 1) return -1;
    Walk w;
    if (walk(*in, w)) return -1;
    std::string out;
    char line[1024];
    if (header)
        out += "# rID\t  gID\t   %id\t  ExonL\t MisMch\t Unpair\t ref_l\t  ref_r\t  tgt_l\t  tgt_r\t eScore\t IntrnL\t "
               "iScore\t Sig3/I\t Sig5/T  # -  X P DiNuc\n";
    for (size_t i = 0; i < w.ex.size(); ++i) {
        const SpdpExonRecord& e = w.ex[i];
        char ends[6] = "  .  ";
        if (e.Iends[0]) { ends[0] = e.Iends[0]; ends[1] = e.Iends[1]; ends[3] = e.Iends[2]; ends[4] = e.Iends[3]; }
        snprintf(line, sizeof line, "%s\t%s\t%7.2f\t%7d\t%7d\t%7d\t%7d\t%7d\t%7d\t%7d\t%7.1f\t%7d\t%7.1f\t%7.2f\t%7.2f %2d %d %2d %d %s\n",
                 qname, gname, e.Pmatch, e.Elen, w.mmc[i], w.unp[i], e.Rleft, e.Rright, e.Gleft, e.Gright,
                 e.Escore, e.Ilen, e.Iscore, e.Sig3, e.Sig5, e.Bmmc, e.Bunp, e.miss, e.phase, ends);
        out += line;
    }
    const float hc = (float) (100. * in->hsp_len / (in->q_right - in->q_left));
    snprintf(line, sizeof line, "@ %s %c ( %d %d ) %s [%d:%d] ( %d %d ) S: %.1f =: %.1f C: %.1f "
             "T#: %d T-: %d B#: %d B-: %d X: %d Nexn: %d HC: %4.1f\n",
             gname, w.g.Csense ? '-' : '+', w.g.Gstart, w.g.Gend, qname, in->q_many, in->q_len, w.g.Rstart, w.g.Rend,
             w.g.Gscore, w.g.Pmatch, w.g.Pcover, w.g.mmc, w.g.unp, w.g.bmmc, w.g.bunp, w.g.ng, (int) w.g.nexn, hc);
    out += line;
    if ((int) out.size() + 1 > cap) return -(int) out.size() - 1;           // the size needed, negated
    memcpy(buf, out.c_str(), out.size() + 1);
    return (int) out.size();
}

// ---- the -O12 record files (src/sqpr.cc:853-885, 960-985): <prefix>.grd = GeneRecord[], <prefix>.erd = ExonRecord[],
//      <prefix>.qrd = the database name, then one query name per gene record, each NUL-terminated.  GeneRecord::Nrecord
//      (exon records written before this gene) and ::Rid (index into .qrd; entry 0 is the database name) are kept here.
struct SpdpO12 {
    FILE* fg = nullptr; FILE* fe = nullptr; FILE* fq = nullptr;
    uint32_t n_exons = 0;
    int32_t n_genes = 0;
};

extern "C" SpdpO12* spdp_o12_open(const char* prefix, const char* db_name)
{
    if (!prefix || !db_name) return nullptr;
    SpdpO12* h = new SpdpO12;
    const std::string p(prefix);
    h->fg = fopen((p + ".grd").c_str(), "wb");
    h->fe = fopen((p + ".erd").c_str(), "wb");
    h->fq = fopen((p + ".qrd").c_str(), "wb");
    if (!h->fg || !h->fe || !h->fq || fputs(db_name, h->fq) == EOF || fputc('\0', h->fq) == EOF) {
        if (h->fg) fclose(h->fg);
        if (h->fe) fclose(h->fe);
        if (h->fq) fclose(h->fq);
        delete h;
        return nullptr;
    }
    return h;
}

extern "C" int spdp_o12_write(SpdpO12* h, const SpdpExonRecord* exons, int n_exons, const SpdpGeneRecord* gene, const char* qname)
{
    if (!h || !gene || !qname || n_exons < 0 || (n_exons && !exons)) return -1;
    SpdpGeneRecord g = *gene;
    g.Nrecord = h->n_exons;
    g.Rid = ++h->n_genes;                       // entry 0 of the .qrd file is the database name
    if (n_exons && fwrite(exons, sizeof(SpdpExonRecord), (size_t) n_exons, h->fe) != (size_t) n_exons) return -1;
    if (fwrite(&g, sizeof g, 1, h->fg) != 1) return -1;
    if (fputs(qname, h->fq) == EOF || fputc('\0', h->fq) == EOF) return -1;
    h->n_exons += (uint32_t) n_exons;
    return 0;
}

extern "C" int spdp_o12_close(SpdpO12* h)
{
    if (!h) return -1;
    const int rc = (fclose(h->fg) | fclose(h->fe) | fclose(h->fq)) ? -1 : 0;
    delete h;
    return rc;
}
