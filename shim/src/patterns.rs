//! The pattern a `FormatType` stands for.  `FormatType::get_pattern` is private in term-guard
//! (constraints/format.rs:217-307), so the strings are restated here, each with the line it comes from; the host
//! layer of this repository holds the same table (term_amd/csrc/host/term_guard.cpp) and its tests pin every one of
//! them against the reference's own vectors (tests/golden/reference_vectors.json, "format_patterns").

/// format.rs:237
pub const EMAIL: &str = r"^[a-zA-Z0-9.!#$%&'*+/=?^_`{|}~-]+@[a-zA-Z0-9](?:[a-zA-Z0-9-]{0,61}[a-zA-Z0-9])?(?:\.[a-zA-Z0-9](?:[a-zA-Z0-9-]{0,61}[a-zA-Z0-9])?)*$";
/// format.rs:241
pub const URL_WITH_LOCALHOST: &str = r"^https?://(?:localhost|(?:[a-zA-Z0-9.-]+\.?[a-zA-Z]{2,}|(?:\d{1,3}\.){3}\d{1,3}))(?::\d+)?(?:/[^\s]*)?$";
/// format.rs:243
pub const URL: &str = r"^https?://[a-zA-Z0-9.-]+\.[a-zA-Z]{2,}(?::\d+)?(?:/[^\s]*)?$";
/// format.rs:248
pub const CREDIT_CARD: &str = r"^(?:4[0-9]{12}(?:[0-9]{3})?|5[1-5][0-9]{14}|3[47][0-9]{13}|3[0-9]{13}|6(?:011|5[0-9]{2})[0-9]{12})$|^(?:\d{4}[-\s]?){3}\d{4}$";
/// format.rs:272
pub const UUID: &str = r"^[0-9a-fA-F]{8}-[0-9a-fA-F]{4}-[1-5][0-9a-fA-F]{3}-[89abAB][0-9a-fA-F]{3}-[0-9a-fA-F]{12}$";
/// format.rs:275
pub const IPV4: &str = r"^(?:(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.){3}(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)$";
/// format.rs:279
pub const IPV6: &str = r"^([0-9a-fA-F]{0,4}:){1,7}([0-9a-fA-F]{0,4})?$|^::$|^::1$|^([0-9a-fA-F]{1,4}:)*::([0-9a-fA-F]{1,4}:)*[0-9a-fA-F]{1,4}$";
/// format.rs:283
pub const JSON: &str = r"^\s*[\{\[].*[\}\]]\s*$";
/// format.rs:287
pub const ISO8601_DATETIME: &str = r"^\d{4}-\d{2}-\d{2}T\d{2}:\d{2}:\d{2}(?:\.\d+)?(?:Z|[+-]\d{2}:\d{2})$";
/// format.rs:294
pub const SSN: &str = r"^(00[1-9]|0[1-9][0-9]|[1-5][0-9]{2}|6[0-5][0-9]|66[0-5]|667|66[89]|6[7-9][0-9]|[7-8][0-9]{2})-?(0[1-9]|[1-9][0-9])-?(000[1-9]|00[1-9][0-9]|0[1-9][0-9]{2}|[1-9][0-9]{3})$";

/// format.rs:252-256
pub fn phone(country: Option<&str>) -> &'static str {
    match country {
        Some("US") | Some("CA") => r"^(\+?1[-.\s]?)?\(?([0-9]{3})\)?[-.\s]?([0-9]{3})[-.\s]?([0-9]{4})$",
        Some("UK") => r"^(\+44\s?)?(?:\(?0\d{4}\)?\s?\d{6}|\(?0\d{3}\)?\s?\d{7}|\(?0\d{2}\)?\s?\d{8})$",
        Some("DE") => r"^(\+49\s?)?(?:\(?0\d{2,5}\)?\s?\d{4,12})$",
        Some("FR") => r"^(\+33\s?)?(?:\(?0\d{1}\)?\s?\d{8})$",
        _ => r"^[\+]?[1-9][\d]{0,15}$",
    }
}
/// format.rs:261-268
pub fn postal_code(country: &str) -> &'static str {
    match country {
        "US" => r"^\d{5}(-\d{4})?$",
        "CA" => r"^[A-Za-z]\d[A-Za-z][ -]?\d[A-Za-z]\d$",
        "UK" => r"^[A-Z]{1,2}\d[A-Z\d]?\s?\d[A-Z]{2}$",
        "DE" | "FR" => r"^\d{5}$",
        "JP" => r"^\d{3}-\d{4}$",
        "AU" => r"^\d{4}$",
        _ => r"^[A-Za-z0-9\s-]{3,10}$",
    }
}

#[cfg(feature = "term-guard")]
pub fn of(format: &term_guard::constraints::FormatType) -> String {
    use term_guard::constraints::FormatType::*;
    match format {
        Regex(p) => p.clone(),
        Email => EMAIL.to_string(),
        Url { allow_localhost } => (if *allow_localhost { URL_WITH_LOCALHOST } else { URL }).to_string(),
        CreditCard { .. } => CREDIT_CARD.to_string(),
        Phone { country } => phone(country.as_deref()).to_string(),
        PostalCode { country } => postal_code(country).to_string(),
        UUID => self::UUID.to_string(),
        IPv4 => IPV4.to_string(),
        IPv6 => IPV6.to_string(),
        Json => JSON.to_string(),
        Iso8601DateTime => ISO8601_DATETIME.to_string(),
        SocialSecurityNumber => SSN.to_string(),
    }
}
