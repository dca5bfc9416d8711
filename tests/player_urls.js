// tests/player_urls.js <uvol.json> [--supports etc2] — test helper (node >= 12): resolves every geometry-frame and texture-segment URL
// of a v2 manifest the way the stock player does (reference src/V2/player.ts:141-174 getGeometryURL / getTextureURL, :207-222
// target selection; src/utils.ts:26-32 isTextureFormatSupported, pad / countHashChar) and prints them as JSON, relative to the
// manifest's directory.  `--supports etc2` stands for a renderer with WEBGL_compressed_texture_etc.
// Written from the reference's documented substitution rules (src/Interfaces.ts:75-132); not a copy of the player.
'use strict'
const fs = require('fs')
const EXT = { mp3: '.mp3', draco: '.drc', ktx2: '.ktx2', etc2: '.etc2' }
const PRIORITY = { ktx2: 0, etc2: 1, etc1: 2 }        // more value => more priority (src/Interfaces.ts TEXTURE_FORMAT_PRIORITY)
const m = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'))
const etc = process.argv.indexOf('--supports') > 0 && process.argv[process.argv.indexOf('--supports') + 1] === 'etc2'
if (m.version !== 'v2') throw new Error('not a v2 manifest')
const hashes = (s) => s.split('').filter((c) => c === '#').length
const zpad = (n, w) => { let t = String(n); while (t.length < w) t = '0' + t; return t }
function resolve(template, inputs, n) {
  const w = hashes(template)
  inputs['[' + '#'.repeat(w) + ']'] = zpad(n, w)
  let p = template
  for (const k of Object.keys(inputs)) p = p.replace(k, inputs[k])    // first occurrence only, like String.replace in the player
  return p
}
// the player asks isTextureFormatSupported(renderer, <target NAME>): 'ktx2' / 'mp4' always, 'etc2' with the extension, anything else no
const supported = (name) => name === 'ktx2' || name === 'mp4' || (name === 'etc2' && etc)
const gTarget = Object.keys(m.geometry.targets)[0]                      // needs targets to be an OBJECT keyed by target name
let tTarget = Object.keys(m.texture.targets)[0]
const sorted = Object.keys(m.texture.targets).sort((a, b) => PRIORITY[m.texture.targets[b].format] - PRIORITY[m.texture.targets[a].format])
for (const t of sorted) if (supported(t)) { tTarget = t; break }
const g = m.geometry.targets[gTarget], t = m.texture.targets[tTarget]
const out = { geometryTarget: gTarget, textureTarget: tTarget, textureFormat: t.format, batchSize: t.sequenceSize, resolution: t.resolution, geometry: [], texture: [] }
for (let i = 0; i < g.frameCount; i++) out.geometry.push(resolve(m.geometry.path, { '[target]': gTarget, '[ext]': EXT[g.format] }, i))
for (let s = 0; s < t.sequenceCount; s++) out.texture.push(resolve(m.texture.path, { '[target]': tTarget, '[type]': 'baseColor', '[tag]': 'default', '[ext]': EXT[t.format] }, s))
console.log(JSON.stringify(out))
