// ref_dump_common.h -- shared by the two translation units of the golden harness
// (TEST INFRASTRUCTURE ONLY; see ref_dump.cc).
#ifndef REF_DUMP_COMMON_H_
#define REF_DUMP_COMMON_H_
#include <vector>
#include <string>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "aln.h"
#include "utilseq.h"
#include "wln.h"
#include "vmf.h"
#include "gsinfo.h"

extern	int	MaxVmfSpace;

// ---------------------------------------------------------------- container
struct Writer {
	FILE*	fd;
	explicit Writer(const char* fn) {
	    fd = fopen(fn, "wb");
	    if (!fd) { perror(fn); exit(1); }
	    fwrite("SPDG1\0\0\0", 1, 8, fd);
	}
	~Writer() { fclose(fd); }
	// dtype: 1 u8, 2 i16, 3 i32, 4 i8
	void put(const char* name, unsigned dtype, const void* p, size_t cnt) {
	    static const int esz[5] = {0, 1, 2, 4, 1};
	    char	nm[32];
	    memset(nm, 0, sizeof(nm));
	    strncpy(nm, name, 31);
	    fwrite(nm, 1, 32, fd);
	    unsigned	hd[2] = {dtype, (unsigned) cnt};
	    fwrite(hd, 4, 2, fd);
	    size_t	nb = cnt * esz[dtype];
	    if (nb) fwrite(p, 1, nb, fd);
	    static const char zero[8] = {0};
	    if (nb % 8) fwrite(zero, 1, 8 - nb % 8, fd);
	}
	void put_i32(const char* name, const std::vector<int>& v) {
	    put(name, 3, v.data(), v.size());
	}
	void put_int(const char* name, int x) { put(name, 3, &x, 1); }
};


inline std::vector<int> skl2vec(const SKL* skl)
{
	std::vector<int>	v;
	if (!skl) return v;
	v.push_back(skl->m);		// flags
	v.push_back(skl->n);		// #corners
	for (int i = 1; i <= skl->n; ++i) {
	    v.push_back(skl[i].m);
	    v.push_back(skl[i].n);
	}
	return v;
}

// same calls, same order as the CLI default set-up (spaln.cc:1471-1494)
inline void set_default_params()	// same calls, same order as the CLI default set-up (spaln.cc:1471-1494)
{
	algmode.lcl = 15;
	alprm.ls = 2;
	algmode.lsg = 1;
	algmode.qck = 3;
	algmode.mlt = 0;
	algmode.mns = 3;
	algmode.thr = 1;
	setalgmode(4, 0);
	setNpam(4, -6);
	setpam(100, 0);			// intra-species PAM (spaln.cc:49)
	setpam(150, 1);			// cross-species PAM (spaln.cc:50)
	setpam(50, WlnPamNo);		// HSP-search PAM (spaln.cc:51)
	setorf(75, 2);			// default ORF length (spaln.cc:52)
	OutPrm.MaxOut = 1;
	OutPrm.SkipLongGap = 1;
	OutPrm.fastanno = 1;
	alprm.scale = 10;
}



// What Wilip (src/wln.cc: the word-lookup HSP search the seeded walks call at the recursion levels, and geneorient / FindHsp
// at level 0 / -1) reads besides the two sequences: the three parameter sets of setwlprm(level) with their reduced
// alphabets, the HSP-search substitution matrix (getSimmtx(WlnPamNo)) and a handful of scalars.  wlparams and hspprm are
// file statics of wln.cc: EndBonus = AvTrc / 2 and RepPen = Vab * 10, DirRep = 20 are rebuilt from their definitions
// (src/wln.cc:37, 145-148); AvrSig of the intron penalty (private) = PenaltyPlus(n) - Penalty(n).
template <class W> inline void dump_wilip_model(W& w, const PwdB* pwd)
{
	std::vector<int> lv, ct, bp;
	for (INT l = 0; l < MaxWlpLevel; ++l) {
	    const WLPRM* p = setwlprm(l);
	    const int bl = p->bitpat? (int) strlen(p->bitpat): 0;
	    const int row[12] = {(int) p->elem, (int) p->tpl, (int) p->mask, (int) p->width, (int) p->gain, (int) p->gain1, (int) p->thr,
		p->xdrp, p->cutoff, (int) p->vthr, bl, (int) bp.size()};
	    lv.insert(lv.end(), row, row + 12);
	    for (int i = 0; i < bl; ++i) bp.push_back(p->bitpat[i] == '1');
	    for (int c = 0; c <= ZZZ; ++c) ct.push_back(p->ConvTab? (int) p->ConvTab[c]: 127);
	}
	w.put_i32("wl_levels", lv);
	w.put_i32("wl_bitpat", bp);
	w.put_i32("wl_convtab", ct);
const	Simmtx*	sm = getSimmtx(WlnPamNo);
const	int	R = sm->rows? sm->rows: sm->dim, C = sm->dim;		// (Simmtx::cols is not set for every matrix; a row has dim entries)
	std::vector<int> mx = {R, C};
	for (int i = 0; i < R; ++i)
	    for (int j = 0; j < C; ++j) mx.push_back((int) sm->mtx[i][j]);
	w.put_i32("wl_mtx", mx);
	int	avrsig = 0;
	for (int n = IntronPrm.llmt; n < IntronPrm.llmt + 64; ++n)
	    if (pwd->IntPen->Penalty(n) > SHRT_MIN) { avrsig = pwd->IntPen->PenaltyPlus(n) - pwd->IntPen->Penalty(n); break; }
	std::vector<int> g = {pwd->DvsP, (int) alprm.scale, (int) ((VTYPE) sm->AvTrc() / 2), (int) ((VTYPE) (alprm.scale * 10.f)), 20,
	    (int) algmode.crs, (int) algmode.lsg, (int) algmode.mlt, (int) IntronPrm.hard_minl, (int) IntronPrm.hard_maxl,
	    IntronPrm.minl, IntronPrm.maxl, shortquery, 0, MET, SER, SER2, ZZZ, avrsig, IntronPrm.llmt, 3};
	w.put_i32("wl_glob", g);
}

int dump_protein(Seq** seqs, const char* exg, const std::vector<int>& udh_list, const char* outfn);
int dump_protein_body(Seq** seqs, PwdB* pwd, const std::vector<int>& udh_list, const char* outfn);

// -B: the reference writes its -O12 record files (<prefix>.grd / .erd / .qrd, Gsinfo::ExonForm with BIN_FORM,
// sqpr.cc:853-985) for the -A0 and the -A2 alignment of the case instead of the -O4 text; the three files end up in the
// fixture byte for byte.  (One process can do one or the other: ExonForm opens its files on its first call only.)
extern int	g_seeded_q;			// -Q n: the seeded path (algmode.qck)
extern std::vector<int>	g_alg_list;		// -A list: the engine selectors of the seeded runs (default 0, 2)
extern bool	wilip_tap_on;			// the Wilip tap of ref_dump.cc
extern std::vector<int>	wilip_tap_log;
extern bool	g_o12_mode;
extern char	g_o12_prefix[256];
inline void o12_begin()
{
	snprintf(g_o12_prefix, sizeof g_o12_prefix, "/tmp/ref_dump_o12_%d", (int) getpid());
	if (!dbs_dt[0]) dbs_dt[0] = new DbsDt('f');
	dbs_dt[0]->dbsid = "fixture_db";		// ExonForm starts the .qrd file with the database name (sqpr.cc:886)
}
inline void o12_write(Gsinfo& gsi, Seq** seqs)
{
	gsi.setprefix(g_o12_prefix);
	gsi.printgene(seqs, BIN_FORM, 0);
}
inline void o12_collect(Writer& w, const Seq* gene)
{
	closeGeneRecord();
const	char*	ext[3] = {".grd", ".erd", ".qrd"};
const	char*	key[3] = {"o12_grd", "o12_erd", "o12_qrd"};
	for (int k = 0; k < 3; ++k) {
	    std::string fn = std::string(g_o12_prefix) + ext[k];
	    std::vector<unsigned char> buf;
	    if (FILE* f = fopen(fn.c_str(), "rb")) {
		int c;
		while ((c = fgetc(f)) != EOF) buf.push_back((unsigned char) c);
		fclose(f);
		remove(fn.c_str());
	    }
	    w.put(key[k], 1, buf.data(), buf.size());
	}
	std::vector<int> meta = {gene->did};
	w.put_i32("o12_meta", meta);
}
#endif
