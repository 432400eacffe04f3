"""Fills the R6_* / R6* placeholders of DESIGN.md sections 5 and 6 from profiles/r6/ (kernel statistics, PMC traffic, the bench line
and detail file, the lanes sweep): the numbers of the round's tables come from the committed evidence, not from memory.
    python tools/fill_design_r6.py [--check]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'profiles', 'r6')


def stats(name):
    out = {}
    with open(os.path.join(P, name + '_kernel_stats.csv')) as f:
        for r in csv.DictReader(f):
            m = re.search(r'opa::(\w+)', r['Name'])
            if m:
                out[m.group(1)] = out.get(m.group(1), 0.0) + float(r['AverageNs']) / 1e3
    return out


def us(v):
    return ('%.1f' % v) if v < 100 else ('%.0f' % v)


pmc = json.load(open(os.path.join(P, 'pmc_traffic.json')))['workloads']
line = json.load(open(os.path.join(P, 'bench_r6.json')))
det = json.load(open(os.path.join(P, 'bench_detail.json')))
s2, s256, sfc, sw, swfc = (stats(n) for n in ('config2', 'config2_b256', 'config2_fc', 'config4', 'config4_fc'))
A32, A256, AW = 200.148608, 1601.188864, 821.071424


def stage(s):
    return sum(v for k, v in s.items() if k.startswith(('cif_', 'cifhr', 'cifseeds', 'cafscored', 'zero')))


def ratio(key):
    w = pmc[key]
    return '%.1f MB = %.2f ×' % (w['decode_path_hbm_bytes'] / 1e6, w['decode_path_hbm_bytes'] / w['algorithmic_bytes'])


rep = {
    'R6_A32': '%.1f' % s2['cifcaf_assoc_kernel'], 'R6_F32': '200.15 MB → %.0f GB/s = **%.4f**' % (A32 / s2['cifcaf_assoc_kernel'] * 1e3, A32 / s2['cifcaf_assoc_kernel'] / 8),
    'R6_D32': '%.1f' % sum(s2.values()), 'R6_CS': us(s2['cafscored_kernel']), 'R6_T': us(s2['cifhr_worktile_kernel']), 'R6_CA': us(s2['cif_active_kernel']),
    'R6_S2': us(s2['cifseeds_sort2k_kernel']), 'R6_S8': us(s2['cifseeds_sort_kernel']), 'R6_RM': us(s2['cifseeds_rankmerge_kernel']),
    'R6_FI': us(s2['cifseeds_fill_cand_kernel']), 'R6_Z': us(s2['zero_kernel']), 'R6_ST': '%.0f' % stage(s2), 'R6_P32': ratio('config2_batch32'),
    'R6_A256': '%.0f' % s256['cifcaf_assoc_kernel'], 'R6_F256': '1 601 MB → %.0f GB/s = **%.3f**' % (A256 / s256['cifcaf_assoc_kernel'] * 1e3, A256 / s256['cifcaf_assoc_kernel'] / 8),
    'R6_D256': '%.0f' % sum(s256.values()), 'R6_CS256': us(s256['cafscored_kernel']), 'R6_T256': us(s256['cifhr_worktile_kernel']),
    'R6_CA256': us(s256['cif_active_kernel']), 'R6_S2256': us(s256['cifseeds_sort2k_kernel']), 'R6_RM256': us(s256['cifseeds_rankmerge_kernel']),
    'R6_FI256': us(s256['cifseeds_fill_cand_kernel']), 'R6_P256': ratio('config2_batch256'),
    'R6_AFC': '%.1f' % sfc['cifcaf_assoc_kernel'], 'R6_FC': '%.1f' % sfc['cifcaf_fc_kernel'], 'R6_CS2': '%.1f' % sfc['cafscored2_kernel'],
    'R6_DFC': '%.3f' % (sum(sfc.values()) / 1e3), 'R6_PFC': ratio('config2_fc_batch32'),
    'R6_AWFC': '%.2f' % (swfc['cifcaf_assoc_kernel'] / 1e3), 'R6_FCW': '%.2f' % (swfc['cifcaf_fc_kernel'] / 1e3), 'R6_CS2W': '%.0f' % swfc['cafscored2_kernel'],
    'R6_DWFC': '%.2f' % (sum(swfc.values()) / 1e3), 'R6_PWFC': ratio('config4_fc_batch16'),
    'R6_AW': '%.2f' % (sw['cifcaf_assoc_kernel'] / 1e3), 'R6_FW': '821.07 MB → %.0f GB/s = **%.4f**' % (AW / sw['cifcaf_assoc_kernel'] * 1e3, AW / sw['cifcaf_assoc_kernel'] / 8),
    'R6_DW': '%.2f' % (sum(sw.values()) / 1e3), 'R6_PW': ratio('config4_batch16'),
    'R6_BENCHA': '%.4f' % line['roofline']['avg_launch_ms'], 'R6_BENCHF': '%.4f' % line['roofline']['frac'],
    'R6_BENCHW': '%.2f' % det['configs']['config4']['roofline']['avg_launch_ms'],
}
ref = det['reference_pipeline']['fp32']
bm = det['roofline']['backbone_mfma']['fp32']
c = det['configs']
lanes = {}
for ln in open(os.path.join(P, 'lanes.log')):
    m = re.match(r'batch (\d+) lanes (\d+) images/s ([\d.]+)', ln)
    if m:
        lanes[(int(m.group(1)), int(m.group(2)))] = float(m.group(3))


def k(v):
    return ('%d' % round(v)).rjust(1)


def thousands(v):
    return '{:,}'.format(int(round(v, -2))).replace(',', ' ')


m4 = lanes[(256, 4)]
rep.update({
    'R6HMS': '%.1f' % line['ms_per_step'], 'R6H': '%.0f' % line['value'], 'R6VB': '%.2f' % line['vs_baseline'], 'R6REF': '%.1f' % ref['images_per_s_1thread'],
    'R6NN': '%.1f' % ref['network_ms_per_batch'], 'R6D2H': '%.0f' % ref['fields_to_host_ms_per_batch'], 'R6CPU': '%.0f' % ref['cpu_decode_ms_per_batch_1thread'],
    'R6DEC': '%.3f' % line['roofline']['decode_path_ms'], 'R6ASSOC': '%.3f' % line['roofline']['avg_launch_ms'], 'R6FRAC': '%.4f' % line['roofline']['frac'],
    'R6PMCX': '%.2f' % (line['roofline']['traffic'] / line['roofline']['algorithmic_bytes_per_launch']), 'R6PMC': '%.1f' % (line['roofline']['traffic'] / 1e6),
    'R6TFF': '%.2f' % bm['frac_of_dense_peak'], 'R6TFD': '%.0f' % bm['TFLOPs_direct_equivalent'], 'R6TF': '%.0f' % bm['TFLOPs'],
    'R6BF': thousands(line['bf16_backbone']['value']).replace('00', '%02d' % (int(line['bf16_backbone']['value']) % 100), 0) if False else '%d' % round(line['bf16_backbone']['value']),
    'R6PREDS': '%.0f' % c['predictor'].get('sync_value', line['configs']['predictor']['sync_value']), 'R6PRED': '%.1f' % line['configs']['predictor']['value'],
    'R6C1REF': '%.1f' % line['configs']['config1_resnet18_321']['ref_cpu_ms'], 'R6C1': '%.2f' % c['config1_resnet18_321']['ms_per_step'],
    'R6C3D': '%.3f' % line['configs']['config3']['decode_ms'], 'R6C3': '%.0f' % c['config3']['value'],
    'R6C4D': '%.2f' % line['configs']['config4']['decode_ms'], 'R6C4': '%.0f' % c['config4']['value'],
    'R6FCMS': '%.2f' % c['force_complete']['ms_per_batch_wall'], 'R6FCV': thousands(c['force_complete']['decode_only_images_per_s']),
    'R6B1G': '%.2f' % c['batch1']['hip_graph_ms_per_image'], 'R6B1': '%.2f' % c['batch1']['eager_ms_per_image'],
    'R6B256F': '%.3f' % c['decode_b256']['roofline']['frac'], 'R6B256': '%.2f' % c['decode_b256']['roofline']['decode_path']['ms_per_batch'],
    'R6L1': thousands(lanes[(32, 1)]), 'R6L2': thousands(lanes[(32, 2)]), 'R6L3': thousands(lanes[(32, 3)]), 'R6L4': thousands(lanes[(32, 4)]),
    'R6M1': thousands(lanes[(256, 1)]), 'R6M2': thousands(lanes[(256, 2)]), 'R6M4GB': '%.0f' % (m4 * 6.2546e-3), 'R6M4F': '%.2f' % (m4 * 6.2546e-3 / 8000), 'R6M4': thousands(m4),
    'R6AAV': thousands(c['all_active']['decode_only_images_per_s']), 'R6AA': '%.1f' % c['all_active']['ms_per_batch_wall'],
})
probe = open(os.path.join(P, 'probe_all_workloads.log')).read()
m = re.search(r'--batch 256 [^\n]*\n(?:config[^\n]*\n)?[^\n]*decode ([\d.]+) ms[^\n]*\nwall: ([\d.]+) ms', probe)
if m:
    rep['R6X_EV'], rep['R6X_WALL'] = m.group(1), m.group(2) + ' = %s images/s' % thousands(256 / float(m.group(2)) * 1e3)
path = os.path.join(ROOT, 'DESIGN.md')
text = open(path).read()
missing = [key for key in rep if key not in text]
for key in sorted(rep, key=len, reverse=True):          # longest first: R6_A256 before R6_A32 before R6_A...
    text = text.replace(key, rep[key])
left = sorted(set(re.findall(r'R6X?_?[A-Z][A-Z0-9_]*', text)))
print('filled %d placeholders; not found in the text: %s; still unfilled: %s' % (len(rep) - len(missing), missing, left))
if '--check' not in sys.argv:
    open(path, 'w').write(text)
